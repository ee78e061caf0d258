"""Build a variant of the library for same-session A/B timing on the GPU box (development tool, never part of the product build).

    python tools/variant_build.py NAME [-DFOO=1 ...]      ->  diffspectra_amd/libdiffspectra_hip_NAME.so
    DIFFSPECTRA_HIP_LIB=diffspectra_amd/libdiffspectra_hip_NAME.so python tools/time_forward.py ...
Only ONE source (default ds_kernels.hip; `--source ds_train_attn.hip` picks another) is recompiled with the extra flags; the other objects come from build/obj (run build() first).
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as g  # noqa: E402


def main():
    name, extra = sys.argv[1], sys.argv[2:]
    which = "ds_kernels.hip"
    if "--source" in extra:
        k = extra.index("--source")
        which = extra[k + 1]
        del extra[k:k + 2]
    src = [s for s in g.HIP_SOURCES if os.path.basename(s) == which][0]
    obj = os.path.join(g.OBJ_DIR, "variant_%s.o" % name)
    others = [os.path.join(g.OBJ_DIR, os.path.basename(s).replace(".hip", ".o")) for s in g.HIP_SOURCES if s != src]
    subprocess.run(["/opt/rocm/bin/hipcc", *g.BASE_FLAGS, *g.OPT_FLAGS, *extra, "-c", src, "-o", obj], check=True, cwd=ROOT)
    out = os.path.join(ROOT, "diffspectra_amd", "libdiffspectra_hip_%s.so" % name)
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", obj, *others, "-o", out], check=True)
    os.remove(obj)
    print(out)


if __name__ == "__main__":
    main()
