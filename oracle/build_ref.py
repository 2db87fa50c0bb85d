"""Builds the one natively-buildable piece of the reference into oracle/_ref/ (git-ignored, travels to the GPU box):
extensions/_cython_bbox/cython_bbox.pyx, cythonized UNMODIFIED from where it lies under /root/reference.
The reference's CUDA (.cu) and TH-API (.c) sources are unbuildable here (no nvcc; TH/THC headers no longer exist in
torch 2.x) -- see DESIGN.md."""
import glob
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/extensions/_cython_bbox/cython_bbox.pyx"
OUT = os.path.join(HERE, "_ref")


def main():
    if not os.path.exists(SRC):
        print("reference not present; keeping whatever is in oracle/_ref/")
        return
    os.makedirs(OUT, exist_ok=True)
    work = tempfile.mkdtemp(prefix="scda_ref_")
    os.symlink(SRC, os.path.join(work, "cython_bbox.pyx"))
    with open(os.path.join(work, "setup_tmp.py"), "w") as f:
        f.write("from setuptools import setup, Extension\nfrom Cython.Build import cythonize\nimport numpy as np\n"
                "setup(ext_modules=cythonize([Extension('cython_bbox', ['cython_bbox.pyx'], include_dirs=[np.get_include()],"
                " extra_compile_args=['-O2'])], language_level=2))\n")
    subprocess.check_call([sys.executable, "setup_tmp.py", "build_ext", "--inplace"], cwd=work,
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    for so in glob.glob(os.path.join(work, "cython_bbox*.so")):
        shutil.copy(so, OUT)
        print("built", os.path.join(OUT, os.path.basename(so)))
    shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
