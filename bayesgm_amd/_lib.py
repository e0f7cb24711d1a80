"""ctypes binding of libbgm_hip.so (C ABI: include/bgm_hip.h).

The product path has no CPU fallback: if the HIP library is missing or a call
fails, a RuntimeError is raised.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BGM_HIP_LIB", os.path.join(_HERE, "libbgm_hip.so"))  # override: dev A/B builds only

BGM_MAX_LAYERS = 8
NET_G, NET_F, NET_H, NET_E = 0, 1, 2, 3
EFFECT_NONE, EFFECT_ADRF, EFFECT_ITE = 0, 1, 2


class CausalConfig(C.Structure):
    _fields_ = [
        ("v_dim", C.c_int32), ("z_dims", C.c_int32 * 4), ("binary_treatment", C.c_int32),
        ("n_hidden_g", C.c_int32), ("g_units", C.c_int32 * BGM_MAX_LAYERS),
        ("n_hidden_f", C.c_int32), ("f_units", C.c_int32 * BGM_MAX_LAYERS),
        ("n_hidden_h", C.c_int32), ("h_units", C.c_int32 * BGM_MAX_LAYERS),
        ("n_hidden_e", C.c_int32), ("e_units", C.c_int32 * BGM_MAX_LAYERS),
        ("sigma_v", C.c_float), ("sigma_x", C.c_float), ("sigma_y", C.c_float),
    ]


class MhArgs(C.Structure):
    _fields_ = [
        ("x_dev", C.c_void_p), ("y_dev", C.c_void_p), ("v_dev", C.c_void_p),
        ("n", C.c_int64), ("row_base", C.c_int64),
        ("state_dev", C.c_void_p), ("logp_dev", C.c_void_p),
        ("init", C.c_int32), ("it_begin", C.c_int32), ("n_iters", C.c_int32), ("burn_in", C.c_int32),
        ("q_sd", C.c_float), ("seed", C.c_uint64),
        ("acc_count_dev", C.c_void_p), ("draws_dev", C.c_void_p),
        ("n_keep", C.c_int32), ("effect", C.c_int32), ("sample_y", C.c_int32),
        ("x_values_dev", C.c_void_p), ("n_doses", C.c_int32),
        ("adrf_partial_dev", C.c_void_p), ("ite_dev", C.c_void_p), ("clock_dev", C.c_void_p),
    ]


class BgmConfig(C.Structure):
    _fields_ = [("x_dim", C.c_int32), ("z_dim", C.c_int32), ("n_hidden_g", C.c_int32),
                ("g_units", C.c_int32 * BGM_MAX_LAYERS)]


class HmcArgs(C.Structure):
    _fields_ = [
        ("x_dev", C.c_void_p), ("n", C.c_int64), ("row_base", C.c_int64),
        ("state_dev", C.c_void_p), ("logp_dev", C.c_void_p), ("grad_dev", C.c_void_p),
        ("init", C.c_int32), ("it_begin", C.c_int32), ("n_iters", C.c_int32), ("burn_in", C.c_int32),
        ("n_leapfrog", C.c_int32), ("step_dev", C.c_void_p), ("seed", C.c_uint64),
        ("acc_prob_sum_dev", C.c_void_p), ("acc_count_dev", C.c_void_p), ("draws_dev", C.c_void_p),
    ]


class BnnConfig(C.Structure):
    _fields_ = [("v_dim", C.c_int32), ("z_dims", C.c_int32 * 4), ("binary_treatment", C.c_int32),
                ("n_hidden", C.c_int32 * 4), ("units", (C.c_int32 * BGM_MAX_LAYERS) * 4),
                ("kl_weight", C.c_float), ("max_batch", C.c_int32), ("norm_mode", C.c_int32),
                ("sigma_v", C.c_float), ("sigma_x", C.c_float), ("sigma_y", C.c_float)]


class BvnConfig(C.Structure):
    _fields_ = [("x_dim", C.c_int32), ("z_dim", C.c_int32), ("n_hidden_g", C.c_int32), ("g_units", C.c_int32 * BGM_MAX_LAYERS),
                ("kl_weight", C.c_float), ("max_batch", C.c_int32), ("hmc_frozen_noise", C.c_int32)]


class BnnMhArgs(C.Structure):
    _fields_ = [("x_dev", C.c_void_p), ("y_dev", C.c_void_p), ("v_dev", C.c_void_p), ("n", C.c_int64), ("row_base", C.c_int64),
                ("block_rows", C.c_int32), ("block0", C.c_int32), ("state_dev", C.c_void_p), ("init", C.c_int32),
                ("it_begin", C.c_int32), ("n_iters", C.c_int32), ("burn_in", C.c_int32), ("q_sd", C.c_float),
                ("seed", C.c_uint64), ("acc_count_dev", C.c_void_p), ("draws_dev", C.c_void_p), ("n_keep", C.c_int32),
                ("effect", C.c_int32), ("sample_y", C.c_int32), ("x_values_dev", C.c_void_p), ("n_doses", C.c_int32),
                ("adrf_sum_dev", C.c_void_p), ("ite_dev", C.c_void_p), ("q_sd_blocks_dev", C.c_void_p), ("acc_blocks_dev", C.c_void_p),
                ("block_row0", C.c_int64)]


class EgmConfig(C.Structure):
    _fields_ = [("batch_size", C.c_int32), ("n_hidden_dz", C.c_int32), ("dz_units", C.c_int32 * BGM_MAX_LAYERS),
                ("lr", C.c_float), ("use_z_rec", C.c_int32)]


class BgmEgmConfig(C.Structure):
    _fields_ = [("batch_size", C.c_int32),
                ("n_hidden_e", C.c_int32), ("e_units", C.c_int32 * BGM_MAX_LAYERS),
                ("n_hidden_dz", C.c_int32), ("dz_units", C.c_int32 * BGM_MAX_LAYERS),
                ("n_hidden_dx", C.c_int32), ("dx_units", C.c_int32 * BGM_MAX_LAYERS),
                ("lr", C.c_float), ("gamma", C.c_float), ("alpha", C.c_float)]


class PriorConfig(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * 5)]


class MhInfo(C.Structure):
    _fields_ = [("rows_per_wave", C.c_int32), ("waves_per_block", C.c_int32), ("grid_blocks", C.c_int32),
                ("mfma_per_transition_per_wave", C.c_int32), ("lds_bytes", C.c_int32),
                ("flop_per_row_transition", C.c_double)]


# every symbol include/bgm_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "bgm_last_error": (C.c_char_p, []),
    "bgm_version": (C.c_char_p, []),
    "bgm_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int]),
    "bgm_set_disc_norm": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_causal_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_bnn_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_causal_set_prior": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "bgm_destroy": (C.c_int, [C.c_void_p]),
    "bgm_causal_configure": (C.c_int, [C.c_void_p, C.POINTER(CausalConfig)]),
    "bgm_causal_set_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_logpost": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_void_p]),
    "bgm_causal_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bgm_causal_mh_slots": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "bgm_causal_mh_run": (C.c_int, [C.c_void_p, C.POINTER(MhArgs), C.c_void_p]),
    "bgm_adrf_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_double,
                                  C.c_void_p, C.c_void_p]),
    "bgm_row_mean_quantiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_double, C.c_double,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_timing_enable": (C.c_int, [C.c_void_p, C.c_int]),
    "bgm_timing_read": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.c_int]),
    "bgm_causal_set_outcome_cache": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_causal_set_event_budget": (C.c_int, [C.c_void_p, C.c_int64]),
    "bgm_causal_outcome_cache_stats": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.c_int32]),
    "bgm_causal_mh_info": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(MhInfo)]),
    "bgm_causal_evaluate": (C.c_int, [C.c_void_p] + [C.c_void_p] * 4 + [C.c_int64, C.c_void_p, C.c_int32,
                                      C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_causal_evaluate_slots": (C.c_int, [C.c_void_p, C.c_int64, C.POINTER(C.c_int32)]),
    "bgm_causal_fit_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "bgm_causal_fit_n_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "bgm_causal_fit_theta_grad": (C.c_int, [C.c_void_p] + [C.c_void_p] * 5 + [C.c_int64, C.c_int32, C.c_int32,
                                            C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_causal_fit_theta_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "bgm_causal_fit_z_step": (C.c_int, [C.c_void_p] + [C.c_void_p] * 7 + [C.c_int64, C.c_int32, C.c_int32, C.c_float,
                                        C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_prior_n_params": (C.c_int, [C.POINTER(PriorConfig), C.POINTER(C.c_int64)]),
    "bgm_prior_table": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_prior_step": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p]),
    "bgm_prior_grad": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_void_p, C.c_float, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_prior_apply": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64,
                                  C.c_void_p]),
    "bgm_bprior_n_params": (C.c_int, [C.POINTER(PriorConfig), C.POINTER(C.c_int64)]),
    "bgm_bprior_step": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_int32, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_int64, C.c_uint64,
                                  C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bprior_apply": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int64,
                                   C.c_void_p]),
    "bgm_bnn_set_prior": (C.c_int, [C.c_void_p, C.POINTER(PriorConfig), C.c_void_p, C.c_void_p]),
    "bgm_causal_describe": (C.c_int, [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]),
    "bgm_causal_fit_z_sync": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p]),
    "bgm_causal_fit_epoch": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_causal_fit_epoch_dp": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_comm_unique_id": (C.c_int, [C.c_void_p]),
    "bgm_comm_create": (C.c_int, [C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_void_p)]),
    "bgm_comm_destroy": (C.c_int, [C.c_void_p]),
    "bgm_comm_info": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "bgm_comm_all_reduce_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_get_weights": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_fit_z_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_causal_fit_state": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int64, C.POINTER(C.c_int64), C.c_void_p]),
    "bgm_causal_fit_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bgm_configure": (C.c_int, [C.c_void_p, C.POINTER(BgmConfig)]),
    "bgm_bgm_fit_begin": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "bgm_bgm_fit_n_params": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    "bgm_bgm_fit_theta_grad": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                         C.c_void_p, C.c_void_p]),
    "bgm_bgm_fit_theta_apply": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p]),
    "bgm_bgm_fit_z_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_float, C.c_void_p,
                                     C.c_void_p]),
    "bgm_bgm_fit_epoch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_void_p,
                                    C.c_void_p]),
    "bgm_bgm_get_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bgm_fit_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bgm_set_weights": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bgm_logpost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bgm_hmc_run": (C.c_int, [C.c_void_p, C.POINTER(HmcArgs), C.c_void_p]),
    "bgm_bgm_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_bvn_set_precision": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_bgm_hmc_adapt": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_double, C.c_float, C.c_float,
                                    C.c_void_p]),
    "bgm_bgm_predict_draws": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_uint64, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                        C.c_void_p]),
    "bgm_causal_effects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                                     C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bgm_fit_set_global_batch": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_bgm_egm_begin": (C.c_int, [C.c_void_p, C.POINTER(BgmEgmConfig), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_void_p]),
    "bgm_bgm_egm_disc_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int32, C.c_void_p,
                                        C.c_void_p]),
    "bgm_bgm_egm_gen_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bgm_egm_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bgm_egm_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bgm_egm_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bgm_bgm_egm_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bgm_egm_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bvn_layout": (C.c_int, [C.POINTER(BvnConfig), C.POINTER(C.c_int64)]),
    "bgm_bvn_begin": (C.c_int, [C.c_void_p, C.POINTER(BvnConfig), C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bvn_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bvn_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bvn_theta_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                     C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bvn_grad_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "bgm_bvn_theta_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "bgm_bvn_z_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_uint64,
                                 C.c_uint32, C.c_void_p, C.c_void_p]),
    "bgm_bvn_logpost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_uint64, C.c_uint32, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "bgm_bvn_hmc_run": (C.c_int, [C.c_void_p, C.POINTER(HmcArgs), C.c_void_p]),
    "bgm_bvn_decode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32, C.c_uint64, C.c_uint32,
                                 C.c_uint32, C.c_uint32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                 C.c_void_p]),
    "bgm_bvn_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bvn_egm_begin": (C.c_int, [C.c_void_p, C.POINTER(BgmEgmConfig), C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p,
                                    C.c_int64, C.c_void_p]),
    "bgm_bvn_egm_disc_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_uint64, C.c_uint32,
                                        C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bvn_egm_gen_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32, C.c_int32,
                                       C.c_void_p, C.c_void_p]),
    "bgm_bvn_egm_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bvn_egm_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bvn_egm_encode": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "bgm_bvn_egm_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bvn_egm_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bnn_begin": (C.c_int, [C.c_void_p, C.POINTER(BnnConfig), C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_layout": (C.c_int, [C.POINTER(BnnConfig), C.POINTER(C.c_int64)]),
    "bgm_bnn_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_write": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_theta_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                     C.c_float, C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bnn_grad_dev": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_int64)]),
    "bgm_bnn_grad_exchange": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "bgm_bnn_theta_apply": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p]),
    "bgm_bnn_z_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                 C.c_int64, C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p,
                                 C.c_void_p, C.c_void_p]),
    "bgm_bnn_z_sync": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_void_p]),
    "bgm_bnn_fit_epoch": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bnn_fit_epoch_dp": (C.c_int, [C.c_void_p] * 8 + [C.c_int64, C.c_int64, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bnn_logpost": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                                  C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p]),
    "bgm_bnn_mh_run": (C.c_int, [C.c_void_p, C.POINTER(BnnMhArgs), C.c_void_p]),
    "bgm_bnn_effects": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_uint64,
                                  C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bnn_evaluate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_void_p,
                                   C.c_int32, C.c_uint64, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "bgm_bnn_egm_begin": (C.c_int, [C.c_void_p, C.POINTER(EgmConfig), C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_egm_disc_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_uint64, C.c_uint32, C.c_int32,
                                        C.c_void_p, C.c_void_p]),
    "bgm_bnn_egm_gen_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint32,
                                       C.c_int32, C.c_void_p, C.c_void_p]),
    "bgm_bnn_egm_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_egm_set_share": (C.c_int, [C.c_void_p, C.c_int32]),
    "bgm_bnn_egm_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_egm_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_bnn_egm_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_bnn_end": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_causal_egm_begin": (C.c_int, [C.c_void_p, C.POINTER(EgmConfig), C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_egm_disc_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_int32, C.c_void_p,
                                           C.c_void_p]),
    "bgm_causal_egm_gen_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_void_p, C.c_void_p]),
    "bgm_causal_egm_read": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_egm_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_egm_apply": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "bgm_causal_egm_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "bgm_causal_egm_end": (C.c_int, [C.c_void_p, C.c_void_p]),
}

_lib = None


def load():
    """Load libbgm_hip.so (raises RuntimeError when it has not been built)."""
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64.  Import torch FIRST so
    # that libbgm_hip.so binds (by SONAME) to the runtime torch already loaded; loading ours first
    # would bring /opt/rocm's copy into the process as a second, device-less HIP runtime.
    import torch  # noqa: F401
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            "bayesgm_amd: %s not found -- build the HIP extension first "
            "(python -m bayesgm_amd.csrc.build); there is no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().bgm_last_error()
        raise RuntimeError("bgm_hip %s failed (%d): %s" % (what, rc, msg.decode() if msg else "?"))
