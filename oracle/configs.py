"""Hyper-parameter sets used by the oracle, the golden-vector generator and the parity tests.

TEST INFRASTRUCTURE ONLY.  Keys are the reference's Optuna parameter names, in the order
the reference's constructors ask for them: FFNN_pre.py:18-38, CNN_pre.py:24-50,
EmbraceNetMultimodal.py:124,134,139-146,156.
"""

# BIOINF_optuna_tuning.db, study A549_active_E_vs_inactive_E_EmbraceNetMultimodal_1augmentation,
# trial 0 -- with every dropout set to 0 (a legal value in each search set) so that train-mode
# parity does not depend on CPU bernoulli masks.
CFG1 = dict(
    FFNN_n_layers=3,
    FFNN_n_units_l0=32, FFNN_dropout_l0=0.0,
    FFNN_n_units_l1=16, FFNN_dropout_l1=0.0,
    FFNN_n_units_l2=16, FFNN_dropout_l2=0.0,
    CNN_n_layers=2,
    CNN_out_channels_l0=64, CNN_kernel_size_l0=15, CNN_dropout_l0=0,
    CNN_out_channels_l1=32, CNN_kernel_size_l1=15, CNN_dropout_l1=0,
    EMBRACENET_embracement_size=512,
    n_post_layers=0,
    selection_probabilities_FFNN=0.5784523087676721,
)
CFG1_F = 48          # A549 epigenomic columns

# two hidden post layers, one-layer pre-nets (d0=64, d1=16*124=1984), c=768
POST2 = dict(
    FFNN_n_layers=1,
    FFNN_n_units_l0=64, FFNN_dropout_l0=0.0,
    CNN_n_layers=1,
    CNN_out_channels_l0=16, CNN_kernel_size_l0=5, CNN_dropout_l0=0,
    EMBRACENET_embracement_size=768,
    n_post_layers=2,
    EMBRACENET_n_units_l0=128, EMBRACENET_dropout_l0=0.0,
    EMBRACENET_n_units_l1=64, EMBRACENET_dropout_l1=0.0,
    selection_probabilities_FFNN=0.3,
)
POST2_F = 58         # H1 epigenomic columns

# small and quick: used for the training-trajectory fixture G9
SMALL = dict(
    FFNN_n_layers=2,
    FFNN_n_units_l0=32, FFNN_dropout_l0=0.0,
    FFNN_n_units_l1=16, FFNN_dropout_l1=0.0,
    CNN_n_layers=2,
    CNN_out_channels_l0=16, CNN_kernel_size_l0=5, CNN_dropout_l0=0,
    CNN_out_channels_l1=32, CNN_kernel_size_l1=11, CNN_dropout_l1=0,
    EMBRACENET_embracement_size=512,
    n_post_layers=1,
    EMBRACENET_n_units_l0=32, EMBRACENET_dropout_l0=0.0,
    selection_probabilities_FFNN=0.42,
)
SMALL_F = 48

CONFIGS = {"cfg1": (CFG1, CFG1_F), "post2": (POST2, POST2_F), "small": (SMALL, SMALL_F)}


class FixedTrial:
    """dict-backed stand-in for an Optuna trial; records the order of suggest_* calls."""

    def __init__(self, params):
        self.params = dict(params)
        self.calls = []

    def _get(self, name):
        self.calls.append(name)
        return self.params[name]

    def suggest_int(self, name, low, high):
        return self._get(name)

    def suggest_categorical(self, name, choices):
        return self._get(name)

    def suggest_float(self, name, low, high):
        return self._get(name)
