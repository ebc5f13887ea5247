#!/usr/bin/env python3
"""
Build a throw-away Python-3 translation of the (Python-2) reference in a scratch
directory OUTSIDE this repository, so that `make_golden.py` can import it and
capture golden input/output vectors.

Recipe: SURVEY.md section 8(c).  Nothing produced here is ever copied into the
repository: only the *data* written by make_golden.py (inputs + expected outputs)
is committed under tests/golden/.  The reference does not exist on the GPU box;
this script is only runnable in the build container where /root/reference is
mounted.

usage: python tests/golden/build_ref.py [scratch_dir]     (default /tmp/segk_ref)
"""
import os
import re
import shutil
import subprocess
import sys

REF = "/root/reference/segmentalist"


def sub(path, pairs):
    with open(path) as f:
        s = f.read()
    for a, b in pairs:
        s, n = re.subn(a, b, s)
    with open(path, "w") as f:
        f.write(s)


def main():
    scratch = sys.argv[1] if len(sys.argv) > 1 else "/tmp/segk_ref"
    if os.path.exists(scratch):
        shutil.rmtree(scratch)
    os.makedirs(scratch)
    pkg = os.path.join(scratch, "segmentalist")
    shutil.copytree(REF, pkg)
    subprocess.check_call(["chmod", "-R", "u+w", scratch])

    # 1. mechanical py2 -> py3
    subprocess.check_call(
        [sys.executable, "-W", "ignore", "-m", "lib2to3", "-w", "-n", "segmentalist"],
        cwd=scratch, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)

    # 2. integer division in index expressions, removed numpy/scipy aliases,
    #    implicit relative import of the Cython sibling
    py_files = []
    for root, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith(".py"):
                py_files.append(os.path.join(root, fn))
    for p in py_files:
        sub(p, [
            (r"\(N\*\*2 \+ N\)/2", "(N**2 + N)//2"),
            (r"\(n_slices\*\*2 \+ n_slices\)/2", "(n_slices**2 + n_slices)//2"),
            (r"t\*\(t - 1\)/2", "t*(t - 1)//2"),
            (r"self\.N_max\*\(self\.N_max \+ 1\)/2", "self.N_max*(self.N_max + 1)//2"),
            (r"i = 0\.5\*\(t - 1\)\*t", "i = int(0.5*(t - 1)*t)"),
            (r"from scipy\.misc import logsumexp", "from scipy.special import logsumexp"),
            (r"np\.float\b(?!\d)", "np.float64"),
            (r"np\.int\b(?!\d)", "np.int64"),
            (r"^import _cython_utils", "from . import _cython_utils"),
            (r"\nimport _cython_utils", "\nfrom . import _cython_utils"),
        ])
    sub(os.path.join(pkg, "tests", "test_bigram_lms.py"), [(r"K = 5\.", "K = 5")])
    sub(os.path.join(pkg, "_cython_utils.pyx"), [(r"np\.int_t", "np.int64_t")])

    # 3. build the Cython extension (language level 2 semantics, as the original)
    import numpy
    env = dict(os.environ)
    env["CFLAGS"] = env.get("CFLAGS", "") + " -I" + numpy.get_include() + " -w"
    subprocess.check_call(["cythonize", "-i", "-2", "segmentalist/_cython_utils.pyx"],
                          cwd=scratch, env=env, stdout=subprocess.DEVNULL)
    print("translated reference ready in", scratch)


if __name__ == "__main__":
    main()
