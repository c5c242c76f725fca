"""Same names as the reference's constants.py:1-6.  Paths are environment-driven here (the
reference hard-codes the authors' machine); MC_DROPOUT_RATE / MC_STEPS keep the reference values."""
import os

DATASET_ROOT = os.environ.get("DASS_DATASET_ROOT", "datasets/")
RUNS = os.environ.get("DASS_RUNS", "runs/")
VISUALIZATIONS_FOLDER = 'visualizations'
TENSORBOARD_VISUALIZATION_INTERVAL = 10  # for every 10% of data
MC_DROPOUT_RATE = 0.25
MC_STEPS = 20
