"""Diagnostic build (-DDS_STAMPS) with only k_attn_fused's in-tree stamps active: wave 0 (a consumer) fills P0-P7, wave 8 (a producer) P8-P15.

    python tools/stamp_attn.py [kernel]   ->  diffspectra_amd/libdiffspectra_hip_stampattn.so (or ..._stamp<kernel>.so)
    DIFFSPECTRA_HIP_LIB=diffspectra_amd/libdiffspectra_hip_stampattn.so python tools/time_forward.py --mols 4096
consumer: P0 prologue | P1 wait for the first tile | P2 logits, P3 barrier | P4 softmax phase | P5 alpha rows + first tile | P6 visits, P7 barrier
producer: P8 prologue | P9 first projection | P10 commit + fetch + projection, P11 barrier | P12 softmax phase | P13 first projection |
          P14 commit + fetch + projection, P15 barrier
Development tool, never part of the product build.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    kernel = sys.argv[1] if len(sys.argv) > 1 else "k_attn_fused"       # any kernel that carries in-tree stamps (k_equi_pairs does)
    s = open(g.HIP_SOURCES[0]).read()
    a = s.index("void %s(Ctx c, int blk) {" % kernel)
    b = s.index("\n}\n", a) + 3
    strip = lambda t: re.sub(r"\n\s*DS_STAMP_FLUSH\([^)]*\);", "", re.sub(r"\n\s*DS_STAMP(_W)?\(\d+\);", "", t)).replace("  DS_STAMP_INIT();\n", "")
    s = strip(s[:a]) + s[a:b] + strip(s[b:])        # the other kernels' stamps share the counters: off
    tmp = os.path.join(ROOT, "diffspectra_amd", "csrc", "_stamped.hip")
    open(tmp, "w").write(s)
    obj = os.path.join(g.OBJ_DIR, "stampattn.o")
    others = [os.path.join(g.OBJ_DIR, os.path.basename(x).replace(".hip", ".o")) for x in g.HIP_SOURCES[1:]]
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", *g.BASE_FLAGS, *g.OPT_FLAGS, "-DDS_STAMPS", "-c", tmp, "-o", obj], check=True, cwd=ROOT)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o",
                        os.path.join(ROOT, "diffspectra_amd", "libdiffspectra_hip_stamp%s.so" % ("attn" if kernel == "k_attn_fused" else kernel))], check=True)
    finally:
        os.remove(tmp)
        if os.path.exists(obj):
            os.remove(obj)


if __name__ == "__main__":
    main()
