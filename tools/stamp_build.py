"""Build the diagnostic (-DDS_STAMPS) library with a phase stamp after every __syncthreads() of the listed kernels.

    python tools/stamp_build.py k_edge_geom:8 k_edge_update:16 k_node_update:24
    DIFFSPECTRA_HIP_LIB=diffspectra_amd/libdiffspectra_hip_stamps.so python tools/time_forward.py ...
Counter index base follows the colon; shares print as P<base+i>.  Development tool, never part of the product build.
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "diffspectra_amd", "csrc", "ds_kernels.hip")


def stamp_kernel(s, name, base):
    a = s.index("void %s(Ctx c" % name)
    a = s.index("{\n", a)
    b = s.index("\n}\n", a)
    k = s[a:b]
    parts = k.split("  __syncthreads();\n")
    out = parts[0]
    if "DS_STAMP_INIT" not in out:
        out = out[:2] + "  DS_STAMP_INIT();\n" + out[2:]
    for i, pt in enumerate(parts[1:]):
        out += "  __syncthreads();\n  DS_STAMP(%d);\n" % (base + i) + pt
    out += "\n  DS_STAMP(%d);\n  DS_STAMP_FLUSH(0);" % (base + len(parts) - 1)
    return s[:a] + out + s[b:], len(parts)


def main():
    s = open(SRC).read()
    import re
    s = re.sub(r"\n\s*DS_STAMP(_W)?\(\d+\);", "", s)
    s = re.sub(r"\n\s*DS_STAMP_FLUSH\([^)]*\);", "", s)           # drop the in-tree stamps, re-insert uniformly
    s = s.replace("  DS_STAMP_INIT();\n", "")
    for spec in sys.argv[1:]:
        name, base = spec.split(":")
        s, n = stamp_kernel(s, name, int(base))
        print(name, "phases:", n, "-> P%d..P%d" % (int(base), int(base) + n - 1))
    tmp = os.path.join(ROOT, "diffspectra_amd", "csrc", "_stamped.hip")
    open(tmp, "w").write(s)
    obj = os.path.join(ROOT, "build", "obj", "stamped.o")
    others = [os.path.join(ROOT, "build", "obj", n) for n in ("ds_aux.o", "ds_train.o")]      # run build() first
    try:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-pass-failed", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
                        "-DDS_STAMPS", "-c", tmp, "-o", obj], check=True)
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o",
                        os.path.join(ROOT, "diffspectra_amd", "libdiffspectra_hip_stamps.so")], check=True)
    finally:
        os.remove(tmp)
        if os.path.exists(obj):
            os.remove(obj)


if __name__ == "__main__":
    main()
