#!/usr/bin/env python3
"""``python train_gmm.py --config gmm_model_config.json [--synthetic | --data-root DIR]`` - the runnable counterpart of the reference's
``trainer_gmm.py`` script on the MI355X HIP path (see music-fader-nets_amd/train.py; the package directory is not a Python identifier,
so it is loaded through mfn_import)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from mfn_import import load_package  # noqa: E402

load_package()
from music_fader_nets_amd.train import main  # noqa: E402

if __name__ == "__main__":
    main()
