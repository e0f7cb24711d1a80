"""Host-side utilities mirrored from the reference's ``bayesgm.utils``.

* ``get_ADRF``  -- utils/helpers.py:8-66 (analytic dose-response truths; the
                   known-answer oracle for ADRF error)
* ``save_data`` -- utils/data_io.py:8-31 (result files written by fit)
* ``parse_file`` / ``parse_file_triplet`` -- utils/data_io.py:33-150 (input files of BGM / CausalBGM)
All pinned by fixtures generated from the reference's own functions (tests/golden/make_golden.py).
"""
import numpy as np


def get_ADRF(x_values=None, x_min=None, x_max=None, nb_intervals=None, dataset='Imbens'):
    valid_datasets = {'Imbens', 'Sun', 'Lee'}
    if dataset not in valid_datasets:
        raise ValueError(f"`dataset` must be one of {valid_datasets}, but got '{dataset}'.")
    if x_values is not None:
        if not isinstance(x_values, (list, np.ndarray)):
            raise ValueError("`x_values` must be a list or numpy array.")
        x_values = np.array(x_values, dtype='float32')
    elif x_min is not None and x_max is not None and nb_intervals is not None:
        if x_min >= x_max:
            raise ValueError("`x_min` must be less than `x_max`.")
        if nb_intervals <= 0:
            raise ValueError("`nb_intervals` must be a positive integer.")
        x_values = np.linspace(x_min, x_max, nb_intervals, dtype='float32')
    else:
        raise ValueError("Either `x_values` or (`x_min`, `x_max`, `nb_intervals`) must be provided.")
    if dataset == 'Imbens':
        return x_values + 2 / (1 + x_values) ** 3
    if dataset == 'Sun':
        return x_values - 0.5 + np.exp(-0.5) + 1
    return 1.2 * x_values + x_values ** 3


def save_data(fname, data, delimiter='\t'):
    if fname.endswith('.npy'):
        np.save(fname, data)
    elif fname.endswith('.txt') or fname.endswith('.csv'):
        np.savetxt(fname, data, fmt='%.6f', delimiter=delimiter)
    else:
        raise ValueError("Wrong saving format, please specify either .npy, .txt, or .csv")


# ---------------------------------------------------------------------------------------------
# Input files (utils/data_io.py:33-150): the data formats on the input side of fit / predict
# ---------------------------------------------------------------------------------------------
def _standardize_columns(a):
    """sklearn StandardScaler().fit_transform semantics: per column (x - mean) / std with the population standard
    deviation computed in float64; columns whose std is (numerically) zero are only centred."""
    a = np.asarray(a)
    a64 = a.astype(np.float64)
    mean = a64.mean(axis=0)
    var = a64.var(axis=0)
    scale = np.sqrt(var)
    eps = 10 * np.finfo(np.float64).eps          # sklearn treats var <= 10 eps * mean^2-ish spreads as constant
    scale[scale < eps * np.maximum(1.0, np.abs(mean))] = 1.0
    scale[var == 0.0] = 1.0
    out = (a64 - mean) / scale
    return out.astype(a.dtype) if a.dtype.kind == 'f' else out


def _read_table(path, sep, header):
    import pandas as pd
    if path.endswith('csv'):
        return pd.read_csv(path, header=header, sep=sep).values
    return np.loadtxt(path, delimiter=sep)


def parse_file(path, sep='\t', header=0, normalize=True):
    """One data matrix from .npz (key 'data' | 'x' | 'X' | first key), .csv (header row `header`) or .txt; float32;
    columns standardised when `normalize` (data_io.py:33-84)."""
    import os
    import sys
    assert os.path.exists(path), f"File not found: {path}"
    if path.endswith('npz'):
        loaded = np.load(path)
        for key in ('data', 'x', 'X'):
            if key in loaded:
                data = loaded[key]
                break
        else:
            data = loaded[list(loaded.keys())[0]]
    elif path.endswith('csv') or path.endswith('txt'):
        data = _read_table(path, sep, header)
    else:
        print('File format not recognized, please use .npz, .csv or .txt as input.')
        sys.exit()
    data = data.astype('float32')
    if normalize:
        data = _standardize_columns(data)
    return data


def parse_file_triplet(path, sep='\t', header=0, normalize=True):
    """(x [n,1], y [n,1], v [n,p]) from .npz (keys x, y, v) or a .csv / .txt table whose first two columns are the
    treatment and the outcome; v standardised when `normalize` (data_io.py:87-150; the .csv branch always takes the
    first row as header, as the reference does)."""
    import os
    import sys
    assert os.path.exists(path)
    if path[-3:] == 'npz':
        data = np.load(path)
        data_x, data_y, data_v = data['x'], data['y'], data['v']
    elif path[-3:] in ('csv', 'txt'):
        data = _read_table(path, sep, 0)
        data_x = data[:, 0].reshape(-1, 1).astype('float32')
        data_y = data[:, 1].reshape(-1, 1).astype('float32')
        data_v = data[:, 2:].astype('float32')
    else:
        print('File format not recognized, please use .npz, .csv or .txt as input.')
        sys.exit()
    if normalize:
        data_v = _standardize_columns(data_v)
    return data_x, data_y, data_v
