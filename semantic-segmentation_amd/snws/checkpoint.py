"""Checkpoint files in the reference's format (utils/saver.py:24-47; written at train.py:272-277 and 327-334, read at
train.py:216-231): a torch.save'd dict {'epoch', 'state_dict', 'optimizer', 'best_pred'}.  The models of this package
keep the reference's parameter / buffer names and shapes, so 'state_dict' entries move both ways unchanged; the
'optimizer' entry is torch.optim.AdamW's state_dict, which snws.optim.FlatAdamW reads and writes too.  Files are
loaded with weights_only=True (nothing in the file is executed)."""
import glob
import os
import shutil

import torch


def save_checkpoint(path, model, optimizer, epoch, best_pred):
    state = {"epoch": int(epoch), "state_dict": {k: v.detach().cpu() for k, v in model.state_dict().items()},
             "optimizer": optimizer.state_dict() if optimizer is not None else None, "best_pred": float(best_pred)}
    torch.save(state, path)
    return state


def load_checkpoint(path, model, optimizer=None, fine_tune=False):
    """train.py:216-231: returns (start_epoch, best_pred); with fine_tune the optimizer state is not restored and the
    start epoch is 0.  Accepts checkpoints saved from a DataParallel-wrapped model ('module.' key prefix) as well."""
    if not os.path.isfile(path):
        raise RuntimeError("=> no checkpoint found at '{}'".format(path))
    ck = torch.load(path, map_location="cpu", weights_only=True)
    sd = ck["state_dict"]
    if sd and all(k.startswith("module.") for k in sd):
        sd = {k[len("module."):]: v for k, v in sd.items()}
    model.load_state_dict(sd)
    from . import ops
    ops.weights_changed()
    if optimizer is not None and not fine_tune and ck.get("optimizer") is not None:
        optimizer.load_state_dict(ck["optimizer"])
    return (0 if fine_tune else int(ck["epoch"])), float(ck["best_pred"])


class Saver(object):
    """Run directories as the reference lays them out: run/<dataset>/<checkname>/experiment_<id>/checkpoint.pth.tar,
    best_pred.txt next to it, and model_best.pth.tar one level up when a run beats every earlier run's best
    (utils/saver.py:11-47)."""

    def __init__(self, dataset, checkname, root="run"):
        self.directory = os.path.join(root, dataset, checkname)
        self.runs = glob.glob(os.path.join(self.directory, "experiment_*"))
        ids = sorted(int(x.split("_")[-1]) for x in self.runs)
        run_id = ids[-1] + 1 if ids else 0
        self.experiment_dir = os.path.join(self.directory, "experiment_{}".format(run_id))
        os.makedirs(self.experiment_dir, exist_ok=True)

    def save_checkpoint(self, model, optimizer, epoch, best_pred, is_best, filename="checkpoint.pth.tar"):
        filename = os.path.join(self.experiment_dir, filename)
        save_checkpoint(filename, model, optimizer, epoch, best_pred)
        if is_best:
            with open(os.path.join(self.experiment_dir, "best_pred.txt"), "w") as f:
                f.write(str(best_pred))
            previous = [0.0]
            for run in self.runs:
                path = os.path.join(run, "best_pred.txt")
                if os.path.exists(path):
                    with open(path, "r") as f:
                        previous.append(float(f.readline()))
            if not self.runs or best_pred > max(previous):
                shutil.copyfile(filename, os.path.join(self.directory, "model_best.pth.tar"))
        return filename
