"""Build a variant of the library for same-session A/B timing on the GPU box (development tool, never part of the product build).

    python tools/variant_build.py NAME [-DFOO=1 ...]      ->  diffspectra_amd/libdiffspectra_hip_NAME.so
    DIFFSPECTRA_HIP_LIB=diffspectra_amd/libdiffspectra_hip_NAME.so python tools/time_forward.py ...
Only ds_kernels.hip is recompiled with the extra flags; the other objects come from build/obj (run build() first).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    obj = os.path.join(g.OBJ_DIR, "variant_%s.o" % name)
    others = [os.path.join(g.OBJ_DIR, os.path.basename(s).replace(".hip", ".o")) for s in g.HIP_SOURCES[1:]]
    subprocess.run(["/opt/rocm/bin/hipcc", *g.BASE_FLAGS, *g.OPT_FLAGS, *extra, "-c", g.HIP_SOURCES[0], "-o", obj], check=True, cwd=ROOT)
    out = os.path.join(ROOT, "diffspectra_amd", "libdiffspectra_hip_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", out], check=True)
    os.remove(obj)
    print(out)


if __name__ == "__main__":
    main()
