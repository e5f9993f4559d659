"""Class-incremental task splits (reference: tasks.py:1-117). Only the per-step class lists are data;
the accessors return the same values as the reference's get_task_labels / get_per_task_classes."""

_COCO_NON_VOC = [0, 8, 10, 11, 13, 14, 15, 22, 23, 24, 25, 27, 28, 31, 32, 33, 34, 35, 36, 37, 38, 39, 40, 41, 42,
                 43, 46, 47, 48, 49, 50, 51, 52, 53, 54, 55, 56, 57, 58, 59, 60, 61, 65, 70, 73, 74, 75, 76, 77, 78,
                 79, 80, 81, 82, 84, 85, 86, 87, 88, 89, 90]
_COCO_VOC = [1, 2, 3, 4, 5, 6, 7, 9, 16, 17, 18, 19, 20, 21, 44, 62, 63, 64, 67, 72]
_COCO_ALL = sorted(set(range(91)) - {12, 26, 29, 30, 45, 66, 68, 69, 71, 83})


def _ranges(*bounds):
    """_ranges(0, 16, 21) -> {0: [0..15], 1: [16..20]}"""
    return {i: list(range(bounds[i], bounds[i + 1])) for i in range(len(bounds) - 1)}


tasks = {
    "coco": {"offline": {0: _COCO_ALL}, "voc": {0: _COCO_NON_VOC, 1: _COCO_VOC}},
    "voc": {
        "offline": _ranges(0, 21),
        "19-1": _ranges(0, 20, 21),
        "19-1b": {0: [c for c in range(21) if c != 5], 1: [5]},
        "15-5": _ranges(0, 16, 21),
        "15-1": _ranges(0, 16, 17, 18, 19, 20, 21),
        "10-5": _ranges(0, 11, 16, 21),
        "10-2": _ranges(0, 11, 13, 15, 17, 19, 21),
        "10-10": _ranges(0, 11, 21),
    },
    "coco-voc": {"voc": {0: _COCO_NON_VOC, 1: _COCO_VOC}},
}


def get_task_list():
    return [t for ds in tasks for t in tasks[ds]]


def _task(dataset, name, step):
    if dataset not in tasks or name not in tasks[dataset]:
        raise NotImplementedError
    td = tasks[dataset][name]
    assert step in td.keys(), f"You should provide a valid step! [{step} is out of range]"
    return td


def get_task_labels(dataset, name, step):
    td = _task(dataset, name, step)
    return list(td[step]), [l for s in range(step) for l in td[s]], f"{dataset}/{name}"


def get_task_dict(dataset, name, step):
    td = _task(dataset, name, step)
    return {s: td[s] for s in range(step + 1)}


def get_per_task_classes(dataset, name, step):
    td = _task(dataset, name, step)
    return [len(td[s]) for s in range(step + 1)]
