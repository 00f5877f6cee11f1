"""`PolyChordOutput` -- what `run_polychord` returns: the numbers of `<root>.stats` and handles on the chain files, with
the attribute and method names of the reference's class (reference pypolychord/output.py:20-235).

The `.stats` file is read by its labels rather than by line number (the engine writes the reference's layout,
read_write.F90:809-910, so either works); getdist / pandas features are available when those packages are."""
import collections
import os
import re

import numpy as np

try:
    import pandas as pd
except ImportError:                     # pragma: no cover
    pd = None


class PolyChordOutput:
    def __init__(self, base_dir, file_root):
        self.base_dir, self.file_root = base_dir, file_root
        self.logZ = self.logZerr = None
        self.logZs, self.logZerrs = [], []
        self.nposterior = self.nequals = self.ndead = self.nlive = self.nlike = None
        self.avnlike, self.avnlikeslice = [], []
        self.means, self.sigmas = [], []              # "Dim No. Mean Sigma" table (posteriors = True)
        num = r"[-+]?(?:\d+\.?\d*|\.\d+)(?:[eE][-+]?\d+)?"
        with open(self.root + ".stats") as f:
            for line in f:
                m = re.match(r"\s*log\(Z(_\d+)?\)\s*=\s*(%s)\s*\+/-\s*(%s)" % (num, num), line)
                if m:
                    if m.group(1) is None:
                        self.logZ, self.logZerr = float(m.group(2)), float(m.group(3))
                    else:
                        self.logZs.append(float(m.group(2))); self.logZerrs.append(float(m.group(3)))
                    continue
                key = line.split(":")[0].strip()
                rest = line.split(":", 1)[1] if ":" in line else ""
                if key in ("nposterior", "nequals", "ndead", "nlive"):
                    setattr(self, key, int(rest.split()[0]))
                elif key == "nlike":
                    try:                              # "*****" when the count overflows the I8 field
                        self.nlike = int(rest.split()[0])
                    except ValueError:
                        self.nlike = None
                elif key == "<nlike>":
                    a, _, b = rest.partition("(")
                    self.avnlike = [float(x) for x in a.split()]
                    self.avnlikeslice = [float(x) for x in b.replace("per slice )", "").split()]
                else:
                    m = re.match(r"\s*(\d+)\s*(%s)\s*\+/-\s*(%s)\s*$" % (num, num), line)
                    if m:
                        self.means.append(float(m.group(2))); self.sigmas.append(float(m.group(3)))
        self.ncluster = len(self.logZs)
        self.pandas = False
        if pd is not None:
            try:
                self._create_pandas_table()
                self.pandas = True
            except (OSError, ValueError):
                pass

    # ---- file names
    @property
    def root(self):
        return os.path.join(self.base_dir, self.file_root)

    def cluster_root(self, i):
        return os.path.join(self.base_dir, "clusters", "%s_%i" % (self.file_root, i))

    @property
    def paramnames_file(self):
        return self.root + ".paramnames"

    def cluster_paramnames_file(self, i):
        return self.cluster_root(i) + ".paramnames"

    # ---- chains
    @property
    def posterior(self):
        """getdist MCSamples of <root>.txt (loglikes there are -2 logL)"""
        import getdist.mcsamples
        return getdist.mcsamples.loadMCSamples(self.root)

    def cluster_posterior(self, i):
        import getdist.mcsamples
        return getdist.mcsamples.loadMCSamples(self.cluster_root(i))

    @property
    def samples(self):
        """pandas table of the equally weighted posterior samples"""
        if self.pandas:
            return self._samples_table
        print("Install pandas for samples functionality")

    @property
    def loglikes(self):
        if self.pandas:
            return np.array(self._samples_table["loglike"])
        print("Install pandas for loglikes functionality")

    # ---- paramnames
    @staticmethod
    def make_paramnames_file(paramnames, filename):
        with open(filename, "w") as f:
            for name, latex in paramnames:
                f.write("%s   %s\n" % (name, latex))

    def make_paramnames_files(self, paramnames):
        self.make_paramnames_file(paramnames, self.paramnames_file)
        for i, _ in enumerate(self.logZs):
            self.make_paramnames_file(paramnames, self.cluster_paramnames_file(i))
        if self.pandas:
            self._create_pandas_table(paramnames=paramnames)

    # ---- tables
    def _create_pandas_table(self, paramnames=None):
        rows = np.atleast_2d(np.loadtxt(self.root + "_equal_weights.txt"))
        names = ["weight", "loglike"]
        names += ["p%d" % i for i in range(rows.shape[1] - 2)] if paramnames is None else [n for n, _ in paramnames]
        self._samples_table = pd.DataFrame(rows, columns=names).astype(float)
        self._samples_table["loglike"] *= -0.5        # the file holds -2 logL

    def _dataframes_for_printing(self):
        z = collections.OrderedDict(("log(Z_%d)" % (i + 1), "%f +/-  %f" % (a, b)) for i, (a, b) in enumerate(zip(self.logZs, self.logZerrs)))
        info = collections.OrderedDict((k, getattr(self, k)) for k in ("ncluster", "nposterior", "nequals", "ndead", "nlive", "nlike"))
        info["<nlike>"] = self.avnlike
        est = collections.OrderedDict()
        for name in self._samples_table.columns[2:]:
            col = np.array(self._samples_table[name])
            est[name] = "%.3E +/- %.3E" % (col.mean(), col.std())
        return [pd.Series({"log(Z)": "%f +/-  %f" % (self.logZ, self.logZerr)}), pd.Series(z), pd.Series(info), pd.Series(est)]

    def __str__(self):
        if self.pandas:
            return ("Global evidence:\n%s\n\nLocal evidences:\n%s\n\nRun-time information:\n%s\n\nParameter estimates:\n%s"
                    % tuple(x.to_string() for x in self._dataframes_for_printing()))
        return self.root

    __repr__ = __str__
