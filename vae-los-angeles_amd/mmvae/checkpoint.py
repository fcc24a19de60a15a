"""Training-state checkpoints for true resume.

The reference only ever saves `model.state_dict()` (train_dna2rna.py:230-231, optimize_hyperparameters.py:209-210):
its runs cannot be resumed.  SURVEY.md section 8(f)-3 asks for the missing half: optimiser moments + step counts, the LR
scheduler, and -- specific to this implementation -- the position of the device-resident Philox stream that feeds the
dropout masks and eps, so that "K steps, save, load, N-K steps" walks the same trajectory as N steps.

The file is a plain dict of tensors / numbers (loads with `torch.load(..., weights_only=True)`):
    {"model": state_dict, "optimizer": FusedAdamW.state_dict() (torch.optim.AdamW layout), "scheduler": ...,
     "noise": {"offset": int}, "extra": {...}}
`model` keeps the reference's key set, so `torch.load(path)["model"]` feeds the reference's own
`load_state_dict` (reconstruct_unmatched.py:66)."""
import torch

from . import engine


def training_state(model, optimizer, scheduler=None, **extra):
    dev = next(model.parameters()).device
    return {"model": model.state_dict(), "optimizer": optimizer.state_dict(),
            "scheduler": None if scheduler is None else scheduler.state_dict(),
            "noise": engine.GLOBAL_NOISE.state_dict(dev), "extra": dict(extra)}


def save_training_state(path, model, optimizer, scheduler=None, **extra):
    torch.save(training_state(model, optimizer, scheduler, **extra), path)


def load_training_state(path_or_state, model, optimizer, scheduler=None):
    """Restores everything `save_training_state` wrote; returns the `extra` dict (epoch, best_val, ...)."""
    st = path_or_state if isinstance(path_or_state, dict) else torch.load(path_or_state, weights_only=True, map_location="cpu")
    model.load_state_dict(st["model"])
    optimizer.load_state_dict(st["optimizer"])
    if scheduler is not None and st.get("scheduler") is not None:
        scheduler.load_state_dict(st["scheduler"])
    if st.get("noise") is not None:
        engine.GLOBAL_NOISE.load_state_dict(st["noise"], next(model.parameters()).device)
    return st.get("extra", {})
