"""The transforms of get_dataset (reference dataset/transform.py) as PLANS: each transform draws its randomness on the host in
the reference's order (so a seeded `random` stream yields the same crops and flips) and records what to do; the pixel work
itself — Pillow-exact resampling, flip, ToTensor, Normalize — runs batched on the device (cl4_input_batch, input.hip).

Same class names and constructor arguments as the reference: Compose :22-58, Resize :61-93, RandomHorizontalFlip :317-340,
ToTensor :419-445, Normalize :452-478, RandomResizedCrop :582-660."""
import math
import random


class Plan:
    """What the device has to do with one decoded photo."""
    __slots__ = ("width", "height", "crop", "out_hw", "flip", "to_tensor", "mean", "std")

    def __init__(self, width, height):
        self.width, self.height = width, height
        self.crop = (0, 0, height, width)      # i, j, h, w
        self.out_hw = (height, width)
        self.flip = False
        self.to_tensor = False
        self.mean = self.std = None

    def desc(self, img_off, lbl_off):
        i, j, h, w = self.crop
        return (img_off, lbl_off, self.height, self.width, i, j, h, w, int(self.flip))


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def plan(self, width, height):
        p = Plan(width, height)
        for t in self.transforms:
            t.apply(p)
        if not p.to_tensor or p.mean is None:
            raise RuntimeError("Compose: the device pipeline needs ToTensor and Normalize in the stack (dataset/__init__.py:14-27)")
        return p

    def __repr__(self):
        return self.__class__.__name__ + "(" + "".join(f"\n    {t}" for t in self.transforms) + "\n)"


class Resize:
    """Smaller edge -> size (int) or exact (h, w); image bilinear, label nearest."""

    def __init__(self, size):
        self.size = size

    def apply(self, p):
        if p.crop != (0, 0, p.height, p.width) or p.flip:
            raise RuntimeError("Resize after a crop / flip is not part of the reference's stacks")
        if isinstance(self.size, int):
            w, h = p.width, p.height
            short, long = (w, h) if w <= h else (h, w)
            new_short, new_long = self.size, int(self.size * long / short)
            p.out_hw = (new_long, new_short) if w <= h else (new_short, new_long)
        else:
            p.out_hw = tuple(self.size)

    def __repr__(self):
        return f"{self.__class__.__name__}(size={self.size})"


class RandomResizedCrop:
    def __init__(self, size, scale=(0.08, 1.0), ratio=(3. / 4., 4. / 3.)):
        self.size = size if isinstance(size, tuple) else (size, size)
        self.scale, self.ratio = scale, ratio

    @staticmethod
    def get_params(width, height, scale, ratio):
        area = width * height
        for _ in range(10):
            target_area = random.uniform(*scale) * area
            log_ratio = (math.log(ratio[0]), math.log(ratio[1]))
            aspect_ratio = math.exp(random.uniform(*log_ratio))
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if w <= width and h <= height:
                i = random.randint(0, height - h)
                j = random.randint(0, width - w)
                return i, j, h, w
        in_ratio = width / height                       # central-crop fallback
        if in_ratio < min(ratio):
            w = width
            h = int(round(w / min(ratio)))
        elif in_ratio > max(ratio):
            h = height
            w = int(round(h * max(ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def apply(self, p):
        p.crop = self.get_params(p.width, p.height, self.scale, self.ratio)
        p.out_hw = self.size

    def __repr__(self):
        return f"{self.__class__.__name__}(size={self.size}, scale={self.scale}, ratio={self.ratio})"


class RandomHorizontalFlip:
    def __init__(self, p=0.5):
        self.p = p

    def apply(self, p):
        if random.random() < self.p:
            p.flip = not p.flip

    def __repr__(self):
        return f"{self.__class__.__name__}(p={self.p})"


class ToTensor:
    def apply(self, p):
        p.to_tensor = True

    def __repr__(self):
        return self.__class__.__name__ + "()"


class Normalize:
    def __init__(self, mean, std):
        self.mean, self.std = tuple(mean), tuple(std)

    def apply(self, p):
        p.mean, p.std = self.mean, self.std

    def __repr__(self):
        return f"{self.__class__.__name__}(mean={self.mean}, std={self.std})"
