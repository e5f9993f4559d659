"""Per-kernel resources of the BUILT library (cl4wsis_amd/libcl4wsis_hip.so): scratch bytes per lane, VGPRs, SGPRs, LDS — read from the
code objects' metadata notes (no recompilation). `python tools/probes/kernel_resources.py [--scratch]`; tests/test_cabi.py uses
`kernel_table()` to hold the library to "no kernel uses scratch" (see csrc/label.hip, seg_pseudo_gt_kernel, for why that matters)."""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "cl4wsis_amd", "libcl4wsis_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib=LIB):
    """The gfx950 ELF images of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fb = os.path.join(td, "fatbin.bin")
        subprocess.run(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fb], check=True)
        data = open(fb, "rb").read()
    out, pos = [], 0
    while True:
        pos = data.find(MAGIC, pos)
        if pos < 0:
            break
        n = struct.unpack_from("<Q", data, pos + 24)[0]
        q = pos + 32
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", data, q)
            triple = data[q + 24:q + 24 + tlen].decode()
            q += 24 + tlen
            if "amdgcn" in triple and size:
                out.append(data[pos + off:pos + off + size])
        pos += len(MAGIC)
    return out


def kernel_table(lib=LIB):
    """[{name, scratch, vgpr, agpr, sgpr, lds, wg}] for every kernel of the library."""
    rows = []
    for img in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(img)
            f.flush()
            txt = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", f.name], capture_output=True, text=True, check=True).stdout
        for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
            blk = ".agpr_count:" + blk
            g = lambda key, d=0: (re.search(r"\.%s:\s+(\S+)" % key, blk) or [None, d])[1]
            rows.append({"name": g("name", "?"), "scratch": int(g("private_segment_fixed_size")), "vgpr": int(g("vgpr_count")),
                         "agpr": int(g("agpr_count")), "sgpr": int(g("sgpr_count")), "lds": int(g("group_segment_fixed_size")),
                         "wg": int(g("max_flat_workgroup_size"))})
    return rows


if __name__ == "__main__":
    rows = kernel_table()
    only = "--scratch" in sys.argv
    for r in sorted(rows, key=lambda r: r["name"]):
        if only and not r["scratch"]:
            continue
        name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip() or r["name"]
        print(f"scratch {r['scratch']:5d}  vgpr {r['vgpr']:3d} agpr {r['agpr']:3d} sgpr {r['sgpr']:3d} lds {r['lds']:6d}  {name[:150]}")
    print(len(rows), "kernels;", sum(1 for r in rows if r["scratch"]), "use scratch")
