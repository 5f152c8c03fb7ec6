"""``train``: the reference's training driver for solvers with trainable parts -- unrolled solvers with learned rho / lambda
schedules, plug-and-play priors whose denoiser weights are trained through the solver (reference dprox/algo/primitives.py:112-205).

Kept from the reference: the call (``train(model=, step_fn=, dataset=, savedir=, epochs=, bs=, lr=, weight_decay=, resume=)``),
``step_fn(batch) -> (gt, inp, pred)``, an MSE loss on (gt, pred), AdamW, one ``last.pth`` per epoch holding
``{model, optimizer, epoch, gstep, psnr, best_psnr}`` and ``resume=<file name in savedir>``.  Not kept: the dataset download (no
network here -- ``dataset`` is any iterable / ``torch.utils.data.Dataset`` / tensor of ground-truth images), the torchlight logger
and progress bar (a plain per-epoch history is returned and written next to the checkpoint), and the RL auto-tuner branch
(``AutoTuneSolver`` is out of scope).  The arithmetic of a step -- forward and backward through the unrolled
iterations -- is the solver's own HIP path (algo/autodiff.py); this module only sequences optimiser steps."""
import json
import math
import os
import warnings

import torch
import torch.nn.functional as F

from .. import _backend as be


def _batches(dataset, bs, shuffle, generator):
    """mini-batches from a tensor [N, ...], a torch Dataset, or any iterable of ready-made batches"""
    if isinstance(dataset, str):
        raise ValueError(f"train(dataset={dataset!r}): named datasets are downloaded by the reference (hf.download_dataset); this "
                         "environment has no network -- pass a tensor of images, a torch.utils.data.Dataset or an iterable of batches")
    if isinstance(dataset, torch.Tensor):
        n = dataset.shape[0]
        order = torch.randperm(n, generator=generator) if shuffle else torch.arange(n)
        for i in range(0, n, bs):
            yield dataset[order[i:i + bs]]
        return
    if isinstance(dataset, torch.utils.data.Dataset):
        yield from torch.utils.data.DataLoader(dataset, batch_size=bs, shuffle=shuffle, generator=generator)
        return
    yield from dataset


class TrainLoop:
    """state of one training run: model, optimiser, counters; ``save`` / ``load`` round-trip everything a resumed run needs"""

    FILE = "last.pth"

    def __init__(self, model, lr=1e-4, weight_decay=1e-3, savedir="saved"):
        self.model, self.savedir = model, str(savedir)
        if not any(p.requires_grad for p in model.parameters()):
            raise ValueError("train: the model has no trainable parameter (specialize(..., learned_params=True), or a denoiser "
                             "with requires_grad weights)")
        # over ALL parameters like the reference (algo/primitives.py:150): frozen ones never get a gradient and are skipped by the
        # step, and the optimizer state of a checkpoint keeps its group size whatever is frozen at resume time
        self.optimizer = torch.optim.AdamW(list(model.parameters()), lr=lr, weight_decay=weight_decay)
        self.epoch, self.gstep, self.best_psnr, self.history = 0, 0, 0.0, []
        os.makedirs(self.savedir, exist_ok=True)

    def save(self, name=FILE, psnr=0.0):
        torch.save({"model": self.model.state_dict(), "optimizer": self.optimizer.state_dict(), "epoch": self.epoch, "gstep": self.gstep,
                    "psnr": psnr, "best_psnr": self.best_psnr}, os.path.join(self.savedir, name))
        with open(os.path.join(self.savedir, "history.json"), "w") as f:
            json.dump(self.history, f)

    def load(self, name):
        ckpt = torch.load(os.path.join(self.savedir, name), map_location="cpu")
        self.model.load_state_dict(ckpt["model"])
        self.optimizer.load_state_dict(ckpt["optimizer"])
        self.epoch, self.gstep, self.best_psnr = ckpt["epoch"] + 1, ckpt["gstep"] + 1, ckpt["best_psnr"]
        hist = os.path.join(self.savedir, "history.json")
        if os.path.exists(hist):
            with open(hist) as f:
                self.history = json.load(f)[:self.epoch]

    def step(self, gt, pred):
        loss = F.mse_loss(gt, pred)
        self.optimizer.zero_grad(set_to_none=True)
        loss.backward()
        self.optimizer.step()
        self.gstep += 1
        lv = float(loss.detach())
        return lv, (10.0 * math.log10(1.0 / lv) if lv > 0 else float("inf"))

    def run_epoch(self, step_fn, batches):
        tot_l, tot_p, n = 0.0, 0.0, 0
        for batch in batches:
            for attempt in (0, 1):
                gt, _inp, pred = step_fn(batch)
                try:
                    lv, pv = self.step(gt, pred)
                    break
                except be.F16RangeError as e:             # a split-f16 backward pass left the binary16 range: the networks concerned
                    if attempt:                            # have fallen back to split-bf16 (be.note_f16_backward) -- the step is repeated
                        raise
                    warnings.warn(f"{e} -- repeating the step", RuntimeWarning, stacklevel=2)
            tot_l, tot_p, n = tot_l + lv, tot_p + pv, n + 1
        rec = {"epoch": self.epoch, "loss": tot_l / max(n, 1), "psnr": tot_p / max(n, 1), "steps": n}
        self.history.append(rec)
        self.best_psnr = max(self.best_psnr, rec["psnr"])
        self.save(self.FILE, rec["psnr"])
        self.epoch += 1
        return rec


def train(solver=None, *, model=None, step_fn=None, dataset=None, savedir="saved", epochs=10, bs=2, lr=1e-4, weight_decay=1e-3,
          resume=None, shuffle=True, seed=0):
    """Trains ``model`` (an ``UnrolledSolver`` / any nn.Module whose forward runs a solver) for ``epochs`` passes over ``dataset``.
    ``step_fn(batch) -> (gt, inp, pred)`` builds the observation from a batch of ground-truth images and runs the solver
    (primitives.py:124-205).  ``solver`` (first positional argument of the reference's ``train``) selects its RL auto-tuner there;
    that specialisation is not part of this backend.  Returns the per-epoch history (also in ``savedir/history.json``)."""
    if solver is not None:
        raise ValueError(f"Training {solver} is not supported yet.")          # (the reference's message for anything but AutoTuneSolver)
    if model is None or step_fn is None or dataset is None:
        raise TypeError("train() needs model=, step_fn= and dataset=")
    loop = TrainLoop(model, lr, weight_decay, savedir)
    if resume:
        loop.load(resume)
    else:
        loop.save()
    gen = torch.Generator()
    while loop.epoch < epochs:
        gen.manual_seed(seed + loop.epoch)                 # the shuffle of an epoch depends on its number only: a resumed run sees the same batches
        loop.run_epoch(step_fn, _batches(dataset, bs, shuffle, gen))
    return loop.history
