"""python -m polychordlite_amd.build -- compile libpolychord_hip.so for gfx950 in-tree, then the CPython
extension `pypolychord/_pypolychord` (g++; the reference's module name and call surface) on top of it."""
import os
import subprocess
import sysconfig

HERE = os.path.dirname(os.path.abspath(__file__))


def build_extension(verbose=True):
    """pypolychord/_pypolychord*.so -- links against ../libpolychord_hip.so through an $ORIGIN rpath"""
    import numpy
    src = os.path.join(HERE, "pypolychord", "_pypolychord_module.cpp")
    out = os.path.join(HERE, "pypolychord", "_pypolychord" + sysconfig.get_config_var("EXT_SUFFIX"))
    lib = os.path.join(HERE, "libpolychord_hip.so")
    if os.path.exists(out) and os.path.getmtime(out) >= max(os.path.getmtime(src), os.path.getmtime(lib)):
        return out
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-I" + sysconfig.get_paths()["include"],
           "-I" + numpy.get_include(), "-I" + os.path.join(os.path.dirname(HERE), "include"), src, "-o", out,
           "-L" + HERE, "-lpolychord_hip", "-ldl", "-Wl,-rpath,$ORIGIN/.."]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return out


def build(verbose=True):
    cmd = ["make", "-C", os.path.join(HERE, "csrc"), "-j4"]
    if verbose:
        print(" ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    build_extension(verbose)
    return os.path.join(HERE, "libpolychord_hip.so")


if __name__ == "__main__":
    print(build())
