"""Stage the reference's OWN, UNMODIFIED Python wrapper for the drop-in tests.

TEST INFRASTRUCTURE.  Packs the `.py` files of /root/reference/python/audioflux, from where they
lie, into oracle/_ref/audioflux_pywrapper.zip -- a build output next to the compiled reference
(oracle/_ref/ is git-ignored, travels to the GPU box like a built .so, is never committed).  No
file of the reference enters the repository; the archive holds the wrapper byte for byte so that
tests/dropin/ can prove the claim "the reference's ctypes wrapper loads libaudioflux_mi355x.so as a
drop-in" on a box where /root/reference does not exist.

Sample audio (utils/sample_data, 3.7 MB) is left out: the drop-in flows use seeded noise.
"""
import os
import sys
import zipfile

HERE = os.path.dirname(os.path.abspath(__file__))
REF_PY = os.environ.get("AFX_REF_PYTHON", "/root/reference/python")
OUT = os.path.join(HERE, "_ref", "audioflux_pywrapper.zip")


def stage(out=OUT, ref_py=REF_PY):
    pkg = os.path.join(ref_py, "audioflux")
    if not os.path.isdir(pkg):
        raise FileNotFoundError(f"reference wrapper not found under {pkg}")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    tmp = out + ".tmp"
    n = 0
    with zipfile.ZipFile(tmp, "w", zipfile.ZIP_DEFLATED) as z:
        for root, dirs, files in os.walk(pkg):
            dirs[:] = sorted(d for d in dirs if d != "__pycache__")
            for f in sorted(files):
                if f.endswith(".py"):
                    full = os.path.join(root, f)
                    z.write(full, os.path.relpath(full, ref_py))
                    n += 1
    os.replace(tmp, out)
    return out, n


if __name__ == "__main__":
    out, n = stage(*(sys.argv[1:2] or [OUT]))
    print(f"staged {n} wrapper modules -> {out}")
