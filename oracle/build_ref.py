"""Builds the natively-buildable pieces of the reference into oracle/_ref/ (git-ignored, travels to the GPU box),
cythonized UNMODIFIED from where they lie under /root/reference, by this recipe (not the reference's setup.py):
    extensions/_cython_bbox/cython_bbox.pyx   bbox_overlaps (IoU without +1)           -> builds with the system python 3.10
    extensions/_cython_bbox/cython_nms.pyx    greedy NMS (+1 IoU, ">=") and soft-NMS    -> builds with the image's OTHER python:
        it declares `np.ndarray[np.int_t, ndim=1]` (cython_nms.pyx:45,48), a COMPILE-time ctypedef that numpy 2.x removed from its
        .pxd, so the system toolchain (numpy 2.2 + Cython 3.2) answers "cython_nms.pyx:45:23: Invalid type".  The image also
        carries /opt/conda/bin/python3.9 with numpy 1.26.4 + Cython 0.29.24 -- the generation of tools the source was written
        for -- and with that interpreter the unmodified file compiles.  The resulting cython_nms.cpython-39-*.so can only be
        imported by python3.9, so it is used by tests/golden/make_golden_nms.py (run with python3.9) to produce the committed
        keep lists of tests/golden/nms_ref.npz; the tests then compare the C oracle and the HIP kernels with those.
The reference's CUDA (.cu) and TH-API (.c) sources are unbuildable here (no nvcc; TH/THC headers no longer exist in
torch 2.x) -- see DESIGN.md."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = "/root/reference/extensions/_cython_bbox"
PY39 = "/opt/conda/bin/python3.9"
MODULES = (("cython_bbox", sys.executable), ("cython_nms", PY39))      # (module, interpreter whose numpy/Cython build it)
OUT = os.path.join(HERE, "_ref")


def main():
    if not all(os.path.exists(os.path.join(SRC_DIR, m + ".pyx")) for m, _ in MODULES):
        print("reference not present; keeping whatever is in oracle/_ref/")
        return
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp(prefix="scda_ref_")
    for m, _ in MODULES:
        os.symlink(os.path.join(SRC_DIR, m + ".pyx"), os.path.join(work, m + ".pyx"))
    with open(os.path.join(work, "setup_tmp.py"), "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy as np\n"
                "import os\nmods = [os.environ['SCDA_REF_MODULE']]\n"
                "setup(ext_modules=cythonize([Extension(m, [m + '.pyx'], include_dirs=[np.get_include()],"
                " extra_compile_args=['-O2']) for m in mods], language_level=2))\n")
    for m, py in MODULES:   # one module per invocation: a module that does not compile must not take the other one down
        if not os.path.exists(py):
            print("NOT built: %s.pyx (%s is not in this image)" % (m, py))
            continue
        env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
        r = subprocess.run([py, "setup_tmp.py", "build_ext", "--inplace"], cwd=work, capture_output=True, text=True,
                           env=dict(env, SCDA_REF_MODULE=m))
        hits = glob.glob(os.path.join(work, m + "*.so"))
        if r.returncode == 0 and hits:
            shutil.copy(hits[0], OUT)
            print("built", os.path.join(OUT, os.path.basename(hits[0])))
        else:
            why = [l for l in (r.stdout + r.stderr).splitlines() if ".pyx:" in l]
            print("NOT built: %s.pyx (%s)" % (m, why[0].strip() if why else "exit %d" % r.returncode))
    shutil.rmtree(work, ignore_errors=True)


def load(name):
    """import oracle/_ref/<name>*.so (None if it has not been built); sets the numpy aliases cython_nms needs"""
    import importlib.util
    import numpy as np
    if not hasattr(np, "int"):
        np.int = int
    if not hasattr(np, "float"):
        np.float = float
    hits = glob.glob(os.path.join(OUT, name + "*.so"))
    if not hits:
        return None
    spec = importlib.util.spec_from_file_location(name, hits[0])
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


if __name__ == "__main__":
    main()
