"""music-fader-nets_amd: MI355X-native GM-VAE (MusicAttrRegGMVAE) training / inference path of Music FaderNets.

The directory name contains '-' (it follows the upstream repo name), so import it through
``mfn_import.load_package()`` at the repo root, which registers it as ``music_fader_nets_amd``.
"""
from . import arith  # noqa: F401
from .gmm_model import MusicAttrRegGMVAE  # noqa: F401
from .trainer import GMVAETrainer, VAETrainer, beta_schedule, convert_to_one_hot  # noqa: F401
from .vae_model import MusicAttrRegVAE  # noqa: F401
from .model_v2 import MusicAttrCVAE, MusicAttrFaderNets, MusicAttrSingleVAE  # noqa: F401
from .trainer_v2 import CVAETrainer, FaderTrainer, GLSRTrainer, SingleVAETrainer  # noqa: F401
from .decode import clean_output, fader_sweep, greedy_decode  # noqa: F401
from .epochs import cpu_state_dict, training_phase, training_phase_v2  # noqa: F401
from .evaluators import GMMNoteEvaluator, GMMRhythmEvaluator, arousal_transfer, run_through_gmm  # noqa: F401

__all__ = ["MusicAttrRegGMVAE", "MusicAttrRegVAE", "MusicAttrSingleVAE", "MusicAttrCVAE", "MusicAttrFaderNets", "SingleVAETrainer", "CVAETrainer",
           "FaderTrainer", "GLSRTrainer", "GMVAETrainer", "VAETrainer", "beta_schedule", "convert_to_one_hot", "clean_output", "fader_sweep",
           "greedy_decode", "training_phase", "training_phase_v2", "cpu_state_dict", "GMMRhythmEvaluator", "GMMNoteEvaluator", "arousal_transfer", "run_through_gmm"]
