"""Training configuration constants (values of reference src/config.py:7-42)."""
import torch


class Config:
    INPUT_DIM_A = 1177
    INPUT_DIM_B = 1211
    LATENT_DIM = 20
    BATCH_SIZE = 32
    NUM_EPOCHS = 200
    LEARNING_RATE = 5e-4
    WEIGHT_DECAY = 1e-5
    BETA_START = 1e-3
    BETA_WARMUP_EPOCHS = 50
    GAMMA = 1.0
    PATIENCE = 15
    LR_SCHEDULER_FACTOR = 0.5
    LR_SCHEDULER_PATIENCE = 5
    CHECKPOINT_DIR = 'checkpoints'
    BEST_MODEL_NAME = 'best_multivae.pt'
    DEVICE = torch.device("cuda" if torch.cuda.is_available() else "cpu")
    TRAIN_TEST_SPLIT = 0.2
    RANDOM_SEED = 42
