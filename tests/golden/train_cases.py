"""Configuration of the short end-to-end training traces (reference vs HIP agents on the CPU test backend)."""
import numpy as np

SEED = 7
CAPQL = dict(net_arch=[16, 16], batch_size=12, learning_starts=20, buffer_size=500, alpha=0.2)
CAPQL_STEPS = 60
MOSAC = dict(net_arch=[16, 16], batch_size=12, learning_starts=20, buffer_size=500)
MOSAC_WEIGHTS = np.array([0.3, 0.7], dtype=np.float32)
MOSAC_STEPS = 50
GPILS_CONT = dict(net_arch=[16, 16], batch_size=8, learning_starts=16, buffer_size=500, gradient_updates=2, per=True)
GPILS_CONT_STEPS = 40
GPILS = dict(net_arch=[16, 16, 16], batch_size=8, learning_starts=16, buffer_size=500, gradient_updates=2, per=True,
             drop_rate=0.0, layer_norm=True, initial_epsilon=0.3, final_epsilon=0.3, target_net_update_freq=7)
GPILS_STEPS = 50
ENVELOPE = dict(net_arch=[16, 16], batch_size=8, num_sample_w=4, learning_starts=20, buffer_size=500, per=True,
                initial_epsilon=0.5, final_epsilon=0.1, epsilon_decay_steps=40, target_net_update_freq=9, gamma=0.95,
                initial_homotopy_lambda=0.2, final_homotopy_lambda=0.8, homotopy_decay_steps=40)
ENVELOPE_STEPS = 60
SACD = dict(net_arch=[16, 16], batch_size=8, learning_starts=14, buffer_size=500, update_frequency=2, target_net_freq=6,
            tau=0.5)
SACD_WEIGHTS = np.array([0.6, 0.4], dtype=np.float32)
SACD_STEPS = 60
GPIPD_DYNA = dict(net_arch=[16, 16, 16], batch_size=8, learning_starts=16, buffer_size=500, gradient_updates=2, per=True,
                  gpi_pd=True, dyna=True, drop_rate=0.0, layer_norm=True, initial_epsilon=0.3, final_epsilon=0.3,
                  target_net_update_freq=7, dynamics_net_arch=[16, 16], dynamics_ensemble_size=3, dynamics_num_elites=2,
                  dynamics_rollout_len=2, dynamics_rollout_starts=24, dynamics_rollout_freq=6, dynamics_rollout_batch_size=20,
                  dynamics_buffer_size=300, dynamics_uncertainty_threshold=8.655, real_ratio=0.5,
                  dynamics_normalize_inputs=True)
GPIPD_DYNA_TRAIN_FREQ = 12
GPIPD_DYNA_FIT = dict(max_epochs=6)    # few epochs: early stopping is a discrete decision that amplifies fp32 round-off
GPIPD_DYNA_STEPS = 44
GPIPD_DYNA_ENV_ID = "mo-mountaincar-like-treasure-v0"     # any id the reference's ModelEnv has a termination rule for
GPIPD_CONT_DYNA = dict(net_arch=[16, 16], batch_size=8, learning_starts=16, buffer_size=500, gradient_updates=2, per=True,
                       dyna=True, dynamics_net_arch=[16, 16], dynamics_train_freq=12, dynamics_rollout_len=2,
                       dynamics_rollout_starts=24, dynamics_rollout_freq=6, dynamics_rollout_batch_size=20,
                       dynamics_buffer_size=300, dynamics_min_uncertainty=3.143, dynamics_real_ratio=0.5)
GPIPD_CONT_DYNA_STEPS = 44
GPIPD_CONT_DYNA_ENV_ID = "mo-mountaincar-like-point-v0"   # terminates imagined roll-outs at x >= 0.45
MORLD = dict(pop_size=3, exchange_every=14, update_passes=2, neighborhood_size=1, sharing_mechanism=[],   # ("transfer" crashes
             # in the reference with MOSAC learners: morld.py:365 reads a learning_rate attribute MOSAC does not have)
             shared_buffer=True, weight_adaptation_method="PSA", gamma=0.97,
             policy_args=dict(net_arch=[16, 16], batch_size=8, learning_starts=6, buffer_size=500))
MORLD_STEPS = 56
MORLD_WEIGHTS = np.array([[0.0, 1.0], [0.5, 0.5], [1.0, 0.0]])     # what equally_spaced_weights is patched to return (pymoo absent)
SUPPORT = [np.array([1.0, 0.0], dtype=np.float32), np.array([0.0, 1.0], dtype=np.float32),
           np.array([0.5, 0.5], dtype=np.float32)]
WEIGHT = np.array([0.4, 0.6], dtype=np.float32)


def reseed(seed=SEED + 1):
    """Align every host RNG stream the training loops consume (called after construction + parameter loading)."""
    import random

    import torch as th

    random.seed(seed)
    np.random.seed(seed)
    th.manual_seed(seed)
