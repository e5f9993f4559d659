"""CPU: the C-ABI library loads and exports every symbol include/cl4wsis.h declares (no compute calls)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "cl4wsis.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cl4_[a-z0-9_]+)\s*\(", src)))


def test_header_symbols_exported():
    from cl4wsis_amd import _hip
    from cl4wsis_amd.build import build_library
    if not os.path.exists(_hip.LIB_PATH):
        build_library()
    lib = ctypes.CDLL(_hip.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_ctypes_prototypes_cover_header():
    from cl4wsis_amd import _hip
    lib = _hip.load()
    assert lib.cl4_abi_version() == 1
    size_fns = {"cl4_conv2d_fwd_workspace_bytes", "cl4_bn_workspace_bytes", "cl4_peak_workspace_bytes", "cl4_label_workspace_bytes", "cl4_p2_loss_workspace_bytes",
                "cl4_pamr_workspace_bytes", "cl4_ngwp_workspace_bytes", "cl4_balanced_ce_workspace_bytes",
                "cl4_bce_ignore_workspace_bytes", "cl4_p1loss_workspace_bytes", "cl4_cam_smooth_workspace_bytes", "cl4_cam_normalize_workspace_bytes", "cl4_abi_version", "cl4_last_error",
                "cl4_input_ksize", "cl4_input_workspace_bytes", "cl4_target_generation_workspace_bytes", "cl4_conv_debug_u_launches", "cl4_conv_debug_s_launches", "cl4_conv_debug_b_launches", "cl4_conv_debug_f_launches", "cl4_kpack_bytes", "cl4_dwconv5_stats_parts",
                "cl4_stem_weight_bytes", "cl4_to_planes_f16_slot_floats", "cl4_bn_bwd_maxima_floats", "cl4_bn_train_bwd_act_workspace_bytes", "cl4_conv2d_fwd_planes_stats_parts"}
    for n in _declared():
        assert n in _hip.PROTOTYPES or n in size_fns, f"{n} has no ctypes prototype"
    assert lib.cl4_label_workspace_bytes(1, 65, 65) > 65 * 65 * 40
    assert lib.cl4_bn_workspace_bytes(128) == 128 * 32 * 2 * 8
    assert lib.cl4_input_ksize(500, 513) == 3 and lib.cl4_input_ksize(1000, 400) == 7      # Pillow: ceil(max(scale, 1)) * 2 + 1
    assert lib.cl4_input_workspace_bytes(2, 65, 65, 100, 3) >= 2 * 100 * 65 * 3


def test_struct_layouts_match_header_field_order():
    """ctypes Structures mirror the header structs field by field."""
    from cl4wsis_amd import _hip
    src = open(os.path.join(ROOT, "include", "cl4wsis.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for cname, st in (("cl4_conv_desc", _hip.ConvDesc), ("cl4_label_io", _hip.LabelIO), ("cl4_p2_loss_io", _hip.P2LossIO),
                      ("cl4_insmap_io", _hip.InsmapIO), ("cl4_sample_desc", _hip.SampleDesc), ("cl4_pack_job", _hip.PackJob), ("cl4_abs_job", _hip.AbsJob)):
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), src, re.S).group(1)
        fields = []
        for decl in body.split(";"):
            decl = decl.strip()
            if not decl:
                continue
            names = re.sub(r"^(const\s+)?[a-z0-9_]+\s*\*?\s*", "", decl, count=1)
            for nm in names.split(","):
                fields.append(nm.strip().lstrip("*").strip())
        assert fields == [f[0] for f in st._fields_], (cname, fields)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from cl4wsis_amd import _hip
    monkeypatch.setattr(_hip, "_lib", None)
    monkeypatch.setattr(_hip, "LIB_PATH", str(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _hip.load()
