"""ctypes binding of libmorl_hip.so (the C ABI declared in include/morl_hip.h).

The product path has NO fallback: if the gfx950 library is missing or fails to load, ``load_library()``
raises.  Tensors cross the boundary as raw device pointers (``tensor.data_ptr()``) plus explicit sizes.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch as th

PKG = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(PKG, "lib", "libmorl_hip.so")

MORL_MAX_LAYERS = 8
MORL_MAX_OBJ = 8
ABI_VERSION = 13


class NetDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("dims", C.c_int32 * (MORL_MAX_LAYERS + 1)), ("obs_dim", C.c_int32),
                ("reward_dim", C.c_int32), ("n_actions", C.c_int32)]


class UpdateCfg(C.Structure):
    _fields_ = [("gamma", C.c_float), ("homotopy_lambda", C.c_float), ("max_grad_norm", C.c_float),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("adam_step", C.c_int32), ("envelope", C.c_int32), ("apply_step", C.c_int32),
                ("main_forward_done", C.c_int32), ("slab_parts", C.c_int32),
                ("per_tree", C.c_void_p), ("per_idx", C.c_void_p), ("per_running_max", C.c_void_p),
                ("per_levels", C.c_int32), ("per_alpha", C.c_float), ("rows_total", C.c_int64),
                ("shard_params_target", C.c_void_p), ("shard_next_obs", C.c_void_p)]


class UpdateOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("loss", "grad_norm", "priority", "target", "pref", "ac", "q_online_next",
                                          "q_target_next", "q_values")]


class StepIO(C.Structure):
    """``morl_step_io``: the persistent argument block of ``morl_envelope_update_n``."""
    _fields_ = [(n, C.c_void_p) for n in ("params_online", "params_target", "grads", "exp_avg", "exp_avg_sq", "tree", "running_max",
                                          "records")] + [("capacity", C.c_int64)] + \
               [(n, C.c_void_p) for n in ("obs", "next_obs", "rewards", "dones", "actions", "idx", "weights")] + \
               [(n, C.c_int32) for n in ("n_levels", "record_floats", "D", "R", "B", "W")] + [("cfg", UpdateCfg)]


class ACDesc(C.Structure):
    _fields_ = [("algo", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32), ("reward_dim", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden", C.c_int32 * MORL_MAX_LAYERS), ("num_q", C.c_int32),
                ("q_layer_norm", C.c_int32), ("q_drop_rate", C.c_float), ("population", C.c_int32),
                ("max_rows", C.c_int32)]


class ACCfg(C.Structure):
    _fields_ = [("gamma", C.c_float), ("tau", C.c_float), ("alpha", C.c_float),
                ("q_lr", C.c_double), ("policy_lr", C.c_double), ("alpha_lr", C.c_double),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("q_step", C.c_int32), ("policy_step", C.c_int32), ("do_policy", C.c_int32),
                ("policy_iters", C.c_int32), ("do_target", C.c_int32), ("autotune", C.c_int32),
                ("target_entropy", C.c_float), ("policy_noise", C.c_float), ("noise_clip", C.c_float),
                ("n_per", C.c_int32), ("dropout_seed", C.c_uint64),
                ("grad_hook", C.c_void_p), ("grad_hook_user", C.c_void_p)]


# int hook(void* user, int which, float* grads, int64_t count, void* stream) -- morl_ac_cfg.grad_hook
GRAD_HOOK = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_void_p)
# morl_allgather_fn / morl_allreduce_fn of morl_comm_init_custom
ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)


AC_STATE_FIELDS = ("q", "q_target", "q_exp_avg", "q_exp_avg_sq", "pol", "pol_exp_avg", "pol_exp_avg_sq", "pol_target",
                   "log_alpha", "log_alpha_exp_avg", "log_alpha_exp_avg_sq", "action_scale", "action_bias", "q_steps",
                   "pol_steps")
AC_BATCH_FIELDS = ("obs", "actions", "rewards", "next_obs", "dones", "w", "eps_next", "eps_pi", "eps_alpha",
                   "drop_masks")
AC_OUT_FIELDS = ("critic_loss", "q_losses", "policy_loss", "alpha_loss", "alpha", "priority", "target_q", "q_grads",
                 "pol_grads")


class ACState(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in AC_STATE_FIELDS]


class ACBatch(C.Structure):
    _fields_ = [("rows", C.c_int32), ("active", C.c_int32)] + [(n, C.c_void_p) for n in AC_BATCH_FIELDS]


class ACOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in AC_OUT_FIELDS]


class GPIDesc(C.Structure):
    _fields_ = [("obs_dim", C.c_int32), ("reward_dim", C.c_int32), ("n_actions", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MORL_MAX_LAYERS), ("num_nets", C.c_int32), ("layer_norm", C.c_int32),
                ("drop_rate", C.c_float), ("max_rows", C.c_int32), ("max_support", C.c_int32)]


class GPICfg(C.Structure):
    _fields_ = [("gamma", C.c_float), ("min_priority", C.c_float), ("max_grad_norm", C.c_float),
                ("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("adam_step", C.c_int32), ("gpi_pd", C.c_int32), ("n_per", C.c_int32), ("apply_step", C.c_int32),
                ("dropout_seed", C.c_uint64)]


class GPIBatch(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("obs", "actions", "rewards", "next_obs", "dones", "w", "sampled_w", "drop_masks")] + \
               [("rows", C.c_int32), ("K", C.c_int32)]


class GPIPer(C.Structure):
    """``morl_gpi_per`` (the prioritised-replay side of ``morl_gpi_update_n_per``)."""
    _fields_ = [(n, C.c_void_p) for n in ("tree", "running_max", "u01", "records", "idx")] + [("capacity", C.c_int64)] + \
               [(n, C.c_int32) for n in ("n_levels", "record_floats", "D", "R", "action_dim", "B", "doubled", "use_gtd")] + \
               [("alpha", C.c_float), ("min_priority", C.c_float)]


GPI_OUT_FIELDS = ("critic_loss", "td_error", "gtd_error", "target_q", "target_q_envelope", "grads", "grad_norm")


class GPIOut(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in GPI_OUT_FIELDS]


class EnsDesc(C.Structure):
    _fields_ = [("input_dim", C.c_int32), ("output_dim", C.c_int32), ("n_hidden", C.c_int32),
                ("hidden", C.c_int32 * MORL_MAX_LAYERS), ("ensemble_size", C.c_int32), ("max_rows", C.c_int32)]


class EnsCfg(C.Structure):
    _fields_ = [("lr", C.c_double), ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double),
                ("adam_step", C.c_int32), ("weight_decay", C.c_float * MORL_MAX_LAYERS)]


_SIGNATURES = {
    # name: (restype, argtypes)
    "morl_last_error": (C.c_char_p, []),
    "morl_abi_version": (C.c_int, []),
    "morl_is_device_build": (C.c_int, []),
    "morl_ctx_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(NetDesc), C.c_int, C.c_int]),
    "morl_ctx_destroy": (C.c_int, [C.c_void_p]),
    "morl_param_count": (C.c_int64, [C.POINTER(NetDesc)]),
    "morl_ctx_set_fused": (C.c_int, [C.c_void_p, C.c_int]),
    "morl_ctx_set_dw_mode": (C.c_int, [C.c_void_p, C.c_int]),
    "morl_ctx_set_timing": (C.c_int, [C.c_void_p, C.c_int]),
    "morl_ctx_read_timing": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "morl_ctx_read_timing_kinds": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    "morl_gather_batch": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int] +
                          [C.c_void_p] * 6 + [C.c_void_p]),
    "morl_sample_gather": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int64, C.c_int, C.c_int,
                                     C.c_int, C.c_int] + [C.c_void_p] * 9 + [C.c_int, C.c_void_p]),
    "morl_envelope_prepare": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_int, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int] + [C.c_void_p] * 9 +
                              [C.c_int, C.c_void_p]),
    "morl_ctx_invalidate_shadows": (C.c_int, [C.c_void_p]),
    "morl_ctx_set_lazy_targets": (C.c_int, [C.c_void_p, C.c_int]),
    "morl_ctx_set_exact_f32": (C.c_int, [C.c_void_p, C.c_int]),
    "morl_ctx_last_step_bf16": (C.c_int, [C.c_void_p]),
    "morl_ctx_backpressure_seconds": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "morl_ctx_lazy_target_rows": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.c_void_p]),
    "morl_ctx_debug_hidden": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "morl_host_device_pointer": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "morl_gather_fields": (C.c_int, [C.c_void_p, C.c_int, C.c_int64, C.c_void_p, C.c_int, C.c_int,
                                     C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_void_p), C.c_void_p]),
    "morl_qnet_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int,
                                    C.c_void_p, C.c_void_p]),
    "morl_envelope_reduce": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_envelope_reduce_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_envelope_greedy_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "morl_envelope_update": (C.c_int, [C.c_void_p] * 12 + [C.c_int, C.c_int, C.POINTER(UpdateCfg),
                                                           C.POINTER(UpdateOut), C.c_void_p]),
    "morl_envelope_update_n": (C.c_int, [C.c_void_p, C.POINTER(StepIO), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_float,
                                         C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_envelope_slabs": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "morl_envelope_main_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]),
    "morl_envelope_slab_online": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                            C.c_void_p, C.c_void_p]),
    "morl_ctx_shard_lazy": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int]),
    "morl_envelope_update_shard": (C.c_int, [C.c_void_p] * 8 + [C.c_int] * 4 + [C.c_void_p, C.c_void_p,
                                                                                 C.POINTER(UpdateCfg), C.POINTER(UpdateOut),
                                                                                 C.c_void_p]),
    "morl_envelope_step_sharded": (C.c_int, [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 8 + [C.c_int] * 4 +
                                   [C.c_void_p, C.c_void_p, C.POINTER(UpdateCfg), C.c_void_p]),
    "morl_envelope_step_batch_sharded": (C.c_int, [C.c_void_p] * 5 + [C.c_int64] + [C.c_void_p] * 8 + [C.c_int] * 4 +
                                         [C.POINTER(UpdateCfg), C.c_void_p]),
    "morl_envelope_rank_step": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(StepIO), C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_comm_unique_id": (C.c_int, [C.c_void_p]),
    "morl_comm_init": (C.c_int, [C.POINTER(C.c_void_p), C.c_void_p, C.c_int, C.c_int]),
    "morl_comm_init_custom": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_comm_ipc_create": (C.c_int, [C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p]),
    "morl_comm_ipc_connect": (C.c_int, [C.c_void_p, C.c_void_p]),
    "morl_comm_check": (C.c_int, [C.c_void_p]),
    "morl_comm_poll": (C.c_int, [C.c_void_p]),
    "morl_comm_destroy": (C.c_int, [C.c_void_p]),
    "morl_comm_size": (C.c_int, [C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "morl_allgather_q_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "morl_comm_wait": (C.c_int, [C.c_void_p, C.c_void_p]),
    "morl_allreduce_grads": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "morl_clip_adam": (C.c_int, [C.c_void_p] * 5 + [C.POINTER(UpdateCfg), C.c_void_p, C.c_void_p]),
    "morl_polyak": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_void_p]),
    "morl_pareto_mask": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]),
    "morl_metrics_workspace_doubles": (C.c_int64, [C.c_int, C.c_int]),
    "morl_hypervolume": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_expected_utility": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                        C.c_void_p]),
    "morl_sumtree_sample": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "morl_sumtree_set": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "morl_sumtree_update": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_void_p,
                                      C.c_void_p, C.c_void_p]),
    "morl_sumtree_update_clamped": (C.c_int, [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_double,
                                              C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_ens_param_count": (C.c_int64, [C.POINTER(EnsDesc)]),
    "morl_ens_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(EnsDesc)]),
    "morl_ens_destroy": (C.c_int, [C.c_void_p]),
    "morl_ens_train_step": (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.POINTER(EnsCfg), C.c_void_p, C.c_void_p]),
    "morl_ens_forward": (C.c_int, [C.c_void_p] * 6 + [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_ens_mse": (C.c_int, [C.c_void_p] * 7 + [C.c_int, C.c_void_p, C.c_void_p]),
    "morl_gpi_param_count": (C.c_int64, [C.POINTER(GPIDesc)]),
    "morl_gpi_mask_bytes": (C.c_int64, [C.POINTER(GPIDesc), C.c_int]),
    "morl_gpi_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(GPIDesc)]),
    "morl_gpi_destroy": (C.c_int, [C.c_void_p]),
    "morl_gpi_update": (C.c_int, [C.c_void_p] * 11 + [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.POINTER(GPICfg),
                                                       C.POINTER(GPIOut), C.c_void_p]),
    "morl_gpi_q_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p]),
    "morl_gpi_action": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                  C.c_void_p, C.c_void_p]),
    "morl_gpi_actions": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                   C.c_void_p]),
    "morl_gpi_actions_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
                                       C.c_void_p, C.c_void_p]),
    "morl_gpi_priorities": (C.c_int, [C.c_void_p] * 8 + [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float,
                                                          C.c_void_p, C.c_void_p]),
    "morl_ac_q_param_count": (C.c_int64, [C.POINTER(ACDesc)]),
    "morl_ac_policy_param_count": (C.c_int64, [C.POINTER(ACDesc)]),
    "morl_ac_mask_bytes": (C.c_int64, [C.POINTER(ACDesc), C.c_int]),
    "morl_ac_create": (C.c_int, [C.POINTER(C.c_void_p), C.POINTER(ACDesc)]),
    "morl_ac_destroy": (C.c_int, [C.c_void_p]),
    "morl_ac_set_gemm_mode": (C.c_int, [C.c_int]),
    "morl_ac_update": (C.c_int, [C.c_void_p, C.POINTER(ACState), C.POINTER(ACBatch), C.POINTER(ACCfg),
                                 C.POINTER(ACOut), C.c_void_p]),
    "morl_ac_update_n": (C.c_int, [C.c_void_p, C.POINTER(ACState), C.c_int, C.POINTER(ACBatch), C.POINTER(ACCfg),
                                   C.POINTER(ACOut), C.c_void_p]),
    "morl_gpi_update_n": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.POINTER(GPIBatch), C.POINTER(GPICfg), C.POINTER(GPIOut),
                                                        C.c_void_p]),
    "morl_gpi_update_n_per": (C.c_int, [C.c_void_p] * 5 + [C.c_int, C.POINTER(GPIPer), C.POINTER(GPIBatch), C.POINTER(GPICfg),
                                        C.POINTER(GPIOut), C.c_void_p]),
    "morl_ac_update_n_per": (C.c_int, [C.c_void_p, C.POINTER(ACState), C.c_int, C.POINTER(GPIPer), C.POINTER(ACBatch), C.POINTER(ACCfg),
                                       C.POINTER(ACOut), C.c_void_p]),
    "morl_ac_policy_forward": (C.c_int, [C.c_void_p, C.POINTER(ACState), C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                         C.c_void_p, C.c_int, C.POINTER(ACCfg), C.c_void_p, C.c_void_p, C.c_void_p]),
    "morl_ac_q_forward": (C.c_int, [C.c_void_p, C.POINTER(ACState), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int,
                                    C.c_int, C.c_void_p, C.c_void_p]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)


_raw_stream = getattr(th._C, "_cuda_getCurrentRawStream", None)
if _raw_stream is None:                                   # (a torch without the private fast path)
    def _raw_stream(index: int) -> int:
        return th.cuda.current_stream(index).cuda_stream


def _ptr(t: Optional[th.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def _chk(t: th.Tensor, dtype, name: str) -> th.Tensor:
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise ValueError(f"{name}: tensor must be contiguous")
    return t


class NativeLib:
    """One loaded instance of the C ABI."""

    def __init__(self, path: str):
        if not os.path.exists(path):
            raise RuntimeError(f"{path} not found: build it with `python morl-baselines_amd/build.py` "
                               "(hipcc --offload-arch=gfx950); there is no CPU fallback")
        self.path = path
        self.lib = C.CDLL(path)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(self.lib, name)  # AttributeError if a declared symbol is missing
            fn.restype = res
            fn.argtypes = args
        if self.lib.morl_abi_version() != ABI_VERSION:
            raise RuntimeError(f"{path}: ABI version {self.lib.morl_abi_version()} != {ABI_VERSION}")
        self.is_device_build = bool(self.lib.morl_is_device_build())

    # -- helpers --------------------------------------------------------------------------------
    def check(self, rc: int) -> None:
        if rc != 0:
            raise RuntimeError(f"libmorl_hip error {rc}: {self.lib.morl_last_error().decode()}")

    def check_device(self, *tensors: Optional[th.Tensor]) -> None:
        """A gfx950 build only accepts device memory; the emulated test build only host memory."""
        for t in tensors:
            if t is None:
                continue
            if self.is_device_build and t.device.type != "cuda":
                raise RuntimeError("libmorl_hip (gfx950 build) was handed a CPU tensor; no CPU fallback exists")
            if not self.is_device_build and t.device.type != "cpu":
                raise RuntimeError("emulated test build was handed a device tensor")

    @staticmethod
    def stream_of(t: th.Tensor) -> int:
        """Raw handle of torch's current stream on the tensor's device (what every entry point enqueues on)."""
        dev = t.device
        if dev.type == "cuda":
            # (the raw-handle query: th.cuda.current_stream() builds a Stream object, ~4 us per call and ten calls per step)
            return _raw_stream(dev.index if dev.index is not None else th.cuda.current_device())
        return 0

    # -- thin wrappers (argument order == header) -----------------------------------------------------
    def ctx_create(self, desc: NetDesc, max_batch: int, max_weights: int) -> int:
        h = C.c_void_p()
        self.check(self.lib.morl_ctx_create(C.byref(h), C.byref(desc), max_batch, max_weights))
        return h.value

    def ctx_destroy(self, h: int) -> None:
        if h:
            self.lib.morl_ctx_destroy(h)

    def param_count(self, desc: NetDesc) -> int:
        return int(self.lib.morl_param_count(C.byref(desc)))


def make_net_desc(obs_dim: int, reward_dim: int, n_actions: int, net_arch: Sequence[int]) -> NetDesc:
    dims = [obs_dim + reward_dim] + [int(h) for h in net_arch] + [n_actions * reward_dim]
    if len(dims) - 1 > MORL_MAX_LAYERS:
        raise ValueError(f"at most {MORL_MAX_LAYERS} linear layers are supported, got {len(dims) - 1}")
    if reward_dim > MORL_MAX_OBJ:
        raise ValueError(f"reward_dim <= {MORL_MAX_OBJ} supported")
    d = NetDesc()
    d.n_layers = len(dims) - 1
    for i, v in enumerate(dims):
        d.dims[i] = v
    d.obs_dim, d.reward_dim, d.n_actions = obs_dim, reward_dim, n_actions
    return d


# Every MORL_* environment variable anything in this repository reads, with its default (INTEGRATION.md "Run-time switches" is the
# same list with the reasons; tests/test_integration_doc.py holds the three -- this table, the document, the getenv sites -- together).
# A MORL_* variable that is NOT here is a typo or a switch of another round: load_library() refuses to start with it set, because
# a silently ignored switch is a benchmark run with the wrong pipeline.
KNOWN_ENV = {
    # library (read once per process by libmorl_hip.so)
    "MORL_EXACT_F32": "0", "MORL_BF_MIN_ROWS": "4096 (weight-sharded rank step: 8192)", "MORL_LAZY_TARGETS": "1",
    "MORL_ARGMAX_IN_CHAIN": "1", "MORL_TD_IN_CHAIN": "1", "MORL_LAZY_MIN_ROWS": "4096 / 8192", "MORL_LAZY_BIG_ROWS": "6144",
    "MORL_AC_LN_CHAIN": "1", "MORL_CHAIN4": "1", "MORL_CHAIN16": "1", "MORL_AC_NMAJOR": "1", "MORL_AC_SCATTER_MAX": "1048576",
    "MORL_AC_ADAM_IN_DW": "1", "MORL_AC_HEADS_PAIRED": "1", "MORL_AC_HEADBWD_IN_CHAIN": "1", "MORL_RCCL_LIB": "",
    "MORL_IPC_TIMEOUT_MS": "3000", "MORL_BFN_TARGETS": "0", "MORL_BFN_MAX_ROWS": "4096", "MORL_BFN_EAGER3": "1", "MORL_PER_SPLIT": "1", "MORL_BF_DUAL": "0", "MORL_BF_T_FIRST": "1", "MORL_BF_ROLL": "0", "MORL_BF_PW": "15",
    # host side
    "MORL_COMM": "rccl", "MORL_HIP_LIB": "morl-baselines_amd/lib/libmorl_hip.so", "MORL_HOST_NOISE": "0",
    # measurement / test infrastructure
    "MORL_BENCH_TIMING": "(by --steps)", "MORL_REFERENCE_ROOT": "/root/reference",
    # build time (morl-baselines_amd/build.py)
    "MORL_BF_PROF": "", "MORL_C16_PROF": "", "MORL_C4_PROF": "",
}


def check_environment(environ=None) -> None:
    """Raise on a ``MORL_*`` environment variable nothing reads (see ``KNOWN_ENV``)."""
    env = os.environ if environ is None else environ
    unknown = sorted(k for k in env if k.startswith("MORL_") and k not in KNOWN_ENV)
    if unknown:
        raise RuntimeError(f"unknown environment variable(s) {', '.join(unknown)}: not a switch of this build "
                           f"(known: {', '.join(sorted(KNOWN_ENV))}; INTEGRATION.md lists what each does)")


_default: Optional[NativeLib] = None


def use_library(lib: Optional[NativeLib]) -> None:
    """Make ``lib`` what ``load_library()`` returns (None: back to the in-tree gfx950 build).  Test hook: objects that
    re-create themselves without an explicit library handle (e.g. an unpickled replay buffer) call ``load_library()``."""
    global _default
    _default = lib


def load_library(path: Optional[str] = None) -> NativeLib:
    """Load (once) the gfx950 library.  Raises if it is absent -- there is deliberately no fallback."""
    global _default
    check_environment()
    if path is not None:
        return NativeLib(path)
    if _default is None:
        _default = NativeLib(os.environ.get("MORL_HIP_LIB", DEFAULT_LIB))
    return _default
