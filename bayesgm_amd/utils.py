"""Host-side helpers either side of the hot path: the analytic dose-response truths the accuracy numbers are measured
against, and the input / result file formats of the command line.

Names and call signatures follow the reference's ``bayesgm.utils`` (utils/helpers.py:8-66, utils/data_io.py:8-150) since
callers import them by name; the code is the build's own.  Behaviour is pinned by fixtures produced with the reference's
functions (tests/golden/make_golden.py, tests/test_datasets_golden.py).
"""
import os

import numpy as np

from .datasets import _column_standardize

# True average dose-response curves of the reference's three continuous-treatment simulations, as functions of a float32
# dose grid (helpers.py:59-64).  'Imbens' is the panel of BASELINE.json's headline config.
_ADRF_TRUTH = {
    "Imbens": lambda t: t + 2 / (1 + t) ** 3,
    "Sun": lambda t: t - 0.5 + np.exp(-0.5) + 1,
    "Lee": lambda t: 1.2 * t + t ** 3,
}


def get_ADRF(x_values=None, x_min=None, x_max=None, nb_intervals=None, dataset='Imbens'):
    """True dose-response curve of a simulated data set on a dose grid given either explicitly (``x_values``: list or
    array) or as ``nb_intervals`` equally spaced doses on [``x_min``, ``x_max``].  float32 grid, as the reference."""
    if dataset not in _ADRF_TRUTH:
        raise ValueError("`dataset` must be one of %s, but got %r." % (sorted(_ADRF_TRUTH), dataset))
    if x_values is not None:
        if not isinstance(x_values, (list, np.ndarray)):
            raise ValueError("`x_values` must be a list or numpy array.")
        grid = np.asarray(x_values, dtype=np.float32)
    elif None not in (x_min, x_max, nb_intervals):
        if not x_min < x_max:
            raise ValueError("`x_min` must be less than `x_max`.")
        if nb_intervals <= 0:
            raise ValueError("`nb_intervals` must be a positive integer.")
        grid = np.linspace(x_min, x_max, nb_intervals, dtype=np.float32)
    else:
        raise ValueError("Either `x_values` or (`x_min`, `x_max`, `nb_intervals`) must be provided.")
    return _ADRF_TRUTH[dataset](grid)


def save_data(fname, data, delimiter='\t'):
    """Result files of fit / the command line: .npy (binary) or .txt / .csv (6 decimals, `delimiter`)."""
    ext = os.path.splitext(fname)[1]
    if ext == '.npy':
        np.save(fname, data)
    elif ext in ('.txt', '.csv'):
        np.savetxt(fname, data, fmt='%.6f', delimiter=delimiter)
    else:
        raise ValueError("Wrong saving format, please specify either .npy, .txt, or .csv")


# ---------------------------------------------------------------------------------------------------------------------
# input files of the command line (the data formats on the input side of fit / predict)
# ---------------------------------------------------------------------------------------------------------------------
def _kind(path):
    """'npz' | 'csv' | 'txt' by the last three characters of the name, else an error exit like the reference's."""
    if not os.path.exists(path):
        raise AssertionError("File not found: %s" % path)
    tail = path[-3:]
    if tail not in ('npz', 'csv', 'txt'):
        raise SystemExit('File format not recognized, please use .npz, .csv or .txt as input.')
    return tail


def _table(path, kind, sep, header):
    """A delimited text table as a 2-D array: .csv through pandas (row `header` holds the column names), .txt through
    numpy (no header)."""
    if kind == 'csv':
        import pandas as pd
        return pd.read_csv(path, header=header, sep=sep).values
    return np.loadtxt(path, delimiter=sep)


def parse_file(path, sep='\t', header=0, normalize=True):
    """One float32 data matrix.  .npz: the array stored under 'data', 'x' or 'X' (in that order of preference), else the
    first array of the archive; .csv / .txt: the whole table.  Columns are standardised when `normalize`."""
    kind = _kind(path)
    if kind == 'npz':
        with np.load(path) as archive:
            key = next((k for k in ('data', 'x', 'X') if k in archive), None) or list(archive.keys())[0]
            mat = archive[key]
    else:
        mat = _table(path, kind, sep, header)
    mat = mat.astype(np.float32)
    return _column_standardize(mat) if normalize else mat


def parse_file_triplet(path, sep='\t', header=0, normalize=True):
    """(x [n,1], y [n,1], v [n,p]).  .npz: arrays 'x', 'y', 'v' as stored; .csv / .txt: first column treatment, second
    outcome, the rest covariates, float32 (a .csv's first row is always taken as the header, whatever `header` says --
    the reference's behaviour).  V is standardised when `normalize`."""
    kind = _kind(path)
    if kind == 'npz':
        with np.load(path) as archive:
            x, y, v = archive['x'], archive['y'], archive['v']
    else:
        tab = _table(path, kind, sep, 0).astype(np.float32)
        x, y, v = tab[:, 0:1], tab[:, 1:2], tab[:, 2:]
    return x, y, (_column_standardize(v) if normalize else v)
