"""Statement coverage of the reference's label-stage code while the label fixtures are generated
(BUILD CONTAINER only — imports /root/reference through ref_harness.py, like make_golden.py).

    python tests/golden/trace_label_coverage.py            # prints + rewrites tests/golden/label_coverage.txt

Runs `make_golden.label_fixture` for every case of LABEL_CASES + LABEL_EDGE_CASES under `sys.settrace` (outputs are written
to a scratch directory, the committed fixtures are not touched) and lists, per traced function of
/root/reference/modules/utils.py, the statement lines that were never executed. The functions are the ones the
phase-2 label stage runs: pseudo_label_generation, refine_label_generation, get_instance_segmentation,
find_instance_center, group_pixels, cluster_peaks, center_map_gen, gaussian.
"""
import dis
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import ref_harness  # noqa: E402

FUNCS = ["pseudo_label_generation", "refine_label_generation", "get_instance_segmentation", "find_instance_center",
         "group_pixels", "cluster_peaks", "center_map_gen", "gaussian"]
# statements that cannot execute on this path, with the reason (kept out of the "missed" list, printed separately)
EXPECTED_UNREACHABLE = {
    "pseudo_label_generation": "lines of the `type(point) is not list` branch (:198-203): train.py:451-461 always passes a list",
    "find_instance_center": "`raise` (:477) needs a batch; the top-k selection (:498-502) needs >= 10000 NMS centres in one contour "
                            "(41x41 NMS leaves < 500 on a 513x513 map)",
    "group_pixels": "`raise` (:517) needs a batch",
    "center_map_gen": "`return center_map` (:103) needs a centre outside the image; centroids and arg-max pixels are inside",
    "get_instance_segmentation": "`except:` body (:593-594): no statement of the try block can raise for these operand types; "
                                 "`else: new_ctr = ctr` (:595-596) needs beta <= 0 (default 3.0); `return fg.long()` (:602) "
                                 "needs ignore=False (validation only, covered by the insmap fixtures)",
}


def _lines(code):
    out = {ln for _, ln in dis.findlinestarts(code) if ln is not None}
    for c in code.co_consts:
        if hasattr(c, "co_code"):
            out |= _lines(c)
    return out


def main():
    R = ref_harness.load_reference()
    import make_golden as mg
    bestie = R["bestie"]
    path = bestie.__file__
    codes = {name: getattr(bestie, name).__code__ for name in FUNCS}
    want = {name: _lines(c) - {c.co_firstlineno} for name, c in codes.items()}
    hit = {name: set() for name in FUNCS}
    spans = {name: (c.co_firstlineno, max(want[name])) for name, c in codes.items()}
    per_case = {}

    def local(frame, event, arg):
        if event == "line":
            ln = frame.f_lineno
            for name, (a, b) in spans.items():
                if a <= ln <= b:
                    hit[name].add(ln)
                    cur.add(ln)
        return local

    def tracer(frame, event, arg):
        if frame.f_code.co_filename == path:
            return local
        return None

    cases = [("main", c) for c in mg.LABEL_CASES] + [("edge", c) for c in mg.LABEL_EDGE_CASES]
    scratch = tempfile.mkdtemp(prefix="cl4_trace_")
    for kind, case in cases:
        cur = set()
        sys.settrace(tracer)
        try:
            if kind == "main":
                mg.label_fixture(R, *case, out_dir=scratch)
            else:
                mg.label_edge_fixture(R, case, out_dir=scratch)
        finally:
            sys.settrace(None)
        per_case[(kind, case)] = cur
    src = open(path).read().splitlines()
    rep = [f"statement coverage of {path} while generating {len(cases)} label fixtures "
           f"({len(mg.LABEL_CASES)} main + {len(mg.LABEL_EDGE_CASES)} edge)"]
    tot_w = tot_h = 0
    for name in FUNCS:
        w, h = want[name], hit[name] & want[name]
        tot_w += len(w)
        tot_h += len(h)
        rep.append(f"{name}: {len(h)}/{len(w)} statements")
        for ln in sorted(w - h):
            rep.append(f"    not executed  :{ln}  {src[ln - 1].strip()}")
        if name in EXPECTED_UNREACHABLE and (w - h):
            rep.append(f"    (expected: {EXPECTED_UNREACHABLE[name]})")
    rep.append(f"total: {tot_h}/{tot_w}")
    rep.append("")
    rep.append("edge statements -> first fixture that executes them")
    for ln in (231, 314, 336, 354, 355, 579, 582, 583, 591, 592, 600):
        first = next((f"{k}:{c}" for (k, c), s in per_case.items() if ln in s), "NONE")
        rep.append(f"    :{ln}  {src[ln - 1].strip()[:70]:70s} {first}")
    text = "\n".join(rep)
    print(text)
    with open(os.path.join(HERE, "label_coverage.txt"), "w") as f:
        f.write(text + "\n")


if __name__ == "__main__":
    main()
