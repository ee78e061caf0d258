"""Diagnostic build: k_attn_fused with a stamp before AND after every barrier (wave 0's work vs. barrier wait per phase).

    python tools/stamp_attn.py [flush_thread]   ->  diffspectra_amd/libdiffspectra_hip_stampattn.so
P0 prologue | phase 1 per chunk: P1 commit, P2 wait, P3 fetch + projection, P4 wait, P5 logits | P6 weights, P7 wait |
P8 V + softmax, P9 wait | phase 2: P10 first fetch/commit (+ wait) | per chunk: P12 fetch + projection, P13 wait, P14 visits, P15 wait
Development tool, never part of the product build.
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402

SRC = g.HIP_SOURCES[0]


def main():
    flush = sys.argv[1] if len(sys.argv) > 1 else "0"
    s = open(SRC).read()
    s = re.sub(r"\n\s*DS_STAMP(_W)?\(\d+\);", "", s)
    s = re.sub(r"\n\s*DS_STAMP_FLUSH\([^)]*\);", "", s)
    s = s.replace("  DS_STAMP_INIT();\n", "")
    a = s.index("void k_attn_fused(Ctx c, int blk) {")
    b = s.index("\n}\n", a)
    k = s[a:b]
    k = k.replace("  ds_fp16_saturate();\n", "  ds_fp16_saturate();\n  DS_STAMP_INIT();\n", 1)

    def rep(old, new, count=1):
        nonlocal k
        assert k.count(old) >= 1, old
        k = k.replace(old, new, count)
    rep("  // ---- phase 1: logits\n", "  DS_STAMP(0);\n  // ---- phase 1: logits\n")
    rep("    commit(false, ck & 1);\n    __syncthreads();", "    commit(false, ck & 1);\n    DS_STAMP(1);\n    __syncthreads();\n    DS_STAMP(2);")
    rep("    project(min(64, P - ck * 64));\n#endif\n    __syncthreads();\n#if !(DS_ABL & 4)", "    project(min(64, P - ck * 64));\n#endif\n    DS_STAMP(3);\n    __syncthreads();\n    DS_STAMP(4);\n#if !(DS_ABL & 4)")
    rep("#endif\n  }\n  load_weights(DS_BW_E1_H);", "#endif\n    DS_STAMP(5);\n  }\n  load_weights(DS_BW_E1_H);")
    rep("  __threadfence_block();\n  __syncthreads();                         // all logits written", "  __threadfence_block();\n  DS_STAMP(6);\n  __syncthreads();\n  DS_STAMP(7);                         // all logits written")
    rep("#endif\n  __threadfence_block();\n  __syncthreads();\n", "#endif\n  __threadfence_block();\n  DS_STAMP(8);\n  __syncthreads();\n  DS_STAMP(9);\n")
    rep("  __syncthreads();\n  for (int ck = 0; ck < nchunks; ++ck) {\n    if (ck + 1 < nchunks) { fetch2_y(ck + 1); fetch2_a(ck + 1); }", "  __syncthreads();\n  DS_STAMP(10);\n  for (int ck = 0; ck < nchunks; ++ck) {\n    if (ck + 1 < nchunks) { fetch2_y(ck + 1); fetch2_a(ck + 1); }")
    rep("    project(min(64, P - ck * 64));\n#endif\n    __syncthreads();\n    if (ck + 1 < nchunks) *reinterpret_cast<uint4*>", "    project(min(64, P - ck * 64));\n#endif\n    DS_STAMP(12);\n    __syncthreads();\n    DS_STAMP(13);\n    if (ck + 1 < nchunks) *reinterpret_cast<uint4*>")
    rep("#endif\n    __syncthreads();                       // Tt / AL are rewritten by the next chunk", "#endif\n    DS_STAMP(14);\n    __syncthreads();\n    DS_STAMP(15);                       // Tt / AL are rewritten by the next chunk")
    k += "\n  DS_STAMP(12);\n  DS_STAMP_FLUSH(%s);" % flush
    s = s[:a] + k + s[b:]
    tmp = os.path.join(ROOT, "diffspectra_amd", "csrc", "_stamped.hip")
    open(tmp, "w").write(s)
    obj = os.path.join(g.OBJ_DIR, "stampattn.o")
    others = [os.path.join(g.OBJ_DIR, os.path.basename(x).replace(".hip", ".o")) for x in g.HIP_SOURCES[1:]]
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", *g.BASE_FLAGS, *g.OPT_FLAGS, "-DDS_STAMPS", "-c", tmp, "-o", obj], check=True, cwd=ROOT)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o",
                        os.path.join(ROOT, "diffspectra_amd", "libdiffspectra_hip_stampattn%s.so" % ("" if flush == "0" else flush))], check=True)
    finally:
        os.remove(tmp)
        if os.path.exists(obj):
            os.remove(obj)


if __name__ == "__main__":
    main()
