"""Host-side utilities mirrored from the reference's ``bayesgm.utils``.

* ``get_ADRF``  -- utils/helpers.py:8-66 (analytic dose-response truths; the
                   known-answer oracle for ADRF error)
* ``save_data`` -- utils/data_io.py:8-31 (result files written by fit)
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
