"""python -m polychordlite_amd.build -- compile libpolychord_hip.so for gfx950 in-tree."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))


def build(verbose=True):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return os.path.join(HERE, "libpolychord_hip.so")


if __name__ == "__main__":
    print(build())
