"""Training loop of the KITTI-masks experiment, drop-in for /root/reference/kitti_masks/solver.py (``Solver``).

Same constructor (``args`` namespace + ``data_loader``), attributes and ``train()`` control flow (:52-96): per batch
``mu = net(x)``, strided split ``mu[::2]`` / ``mu[1::2]``, ``z3 = roll(z1)``, ``LpSimCLRLoss(p, tau=1, compat)``,
``zero_grad / backward / step``, ``log.csv`` running loss, ``last`` / 50 000-iteration checkpoints in the reference's
``{iter, model_states, optim_states}`` layout.  The loss, the encoder's Linear + Softclip tail and Adam (one launch over a
flat arena) are HIP; the convolutions are PyTorch-ROCm.

Data parallel (BASELINE.json configs[4]: "DDP over 4 x MI355X"; the reference itself has none): with an initialised
``torch.distributed`` group of world > 1 every rank encodes its own batch, the negatives pool is the autograd-aware
all-gather of all ranks' ``z1_rec`` (``distributed.gather_negatives``: all-gather forward, reduce-scatter backward) and the
flat gradient arena is all-reduced once per step -- the single-process loss on the concatenated batch.
"""
from __future__ import annotations

import os
import shutil

import torch
import torch.distributed as dist

from .. import losses
from ..distributed import gather_negatives
from ..optim import Adam
from .model import BetaVAE_H as BetaVAE

__all__ = ["Solver"]


class Solver(object):
    def __init__(self, args, data_loader=None):
        self.ckpt_dir = args.ckpt_dir
        self.output_dir = args.output_dir
        self.data_loader = data_loader
        self.dataset = args.dataset
        if not (torch.cuda.is_available() and args.cuda):
            raise RuntimeError("cl_ica_amd.kitti_masks.Solver runs on the GPU only (args.cuda and a visible MI355X): "
                               "the loss / head / optimizer have no CPU path")
        self.device = torch.device("cuda", torch.cuda.current_device())
        self.max_iter = args.max_iter
        self.global_iter = 0

        self.z_dim = args.z_dim
        self.nc = args.num_channel

        # for adam
        self.lr = args.lr
        self.beta1 = args.beta1
        self.beta2 = args.beta2

        self.net = BetaVAE(self.z_dim, self.nc, args.box_norm).to(self.device)
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        if self.world > 1:          # identical replicas
            for prm in self.net.parameters():
                dist.broadcast(prm.data, src=0)
        self.optim = Adam(self.net.parameters(), lr=self.lr, betas=(self.beta1, self.beta2))

        self.ckpt_name = args.ckpt_name
        self.log_step = args.log_step
        self.save_step = args.save_step

        self.loss = losses.LpSimCLRLoss(p=args.p, tau=1.0, simclr_compatibility_mode=True)

    def train_iteration(self, x):
        """The body of the reference's loop for one batch (:61-74); returns the 0-dim loss tensor (no host sync)."""
        x = x.to(self.device)
        mu = self.net(x)
        z1_rec = mu[::2]
        z2_con_z1_rec = mu[1::2]
        if self.world > 1:
            z3_rec = gather_negatives(z1_rec.contiguous())        # all ranks' z1_rec; the row-wise LSE cannot see the order
        else:
            z3_rec = torch.roll(z1_rec, 1, 0)
        vae_loss, _, _ = self.loss(None, None, None, z1_rec, z2_con_z1_rec, z3_rec)
        self.optim.zero_grad()
        vae_loss.backward()
        self.optim.all_reduce_grads()
        self.optim.step()
        return vae_loss

    def train(self):
        self.net_mode(train=True)
        out = False  # whether to exit training loop
        failure = False  # whether training was stopped
        running_loss = 0
        log = open(os.path.join(self.output_dir, "log.csv"), "a", 1)
        log.write("Total Loss\n")

        while not out:
            for x, _ in self.data_loader:  # don't use label
                vae_loss = self.train_iteration(x)
                running_loss += vae_loss.item()

                self.global_iter += 1
                if self.global_iter % self.log_step == 0:
                    running_loss /= self.log_step
                    log.write("%.6f" % running_loss + "\n")
                    running_loss = 0

                if self.global_iter % self.save_step == 0:
                    self.save_checkpoint("last")

                if self.global_iter % 50000 == 0:
                    self.save_checkpoint(str(self.global_iter))

                if self.global_iter >= self.max_iter:
                    out = True
                    break

        if failure:
            shutil.rmtree(self.ckpt_dir)

        return failure

    def save_checkpoint(self, filename, silent=True):
        states = {"iter": self.global_iter,
                  "model_states": {"net": self.net.state_dict()},
                  "optim_states": {"optim": self.optim.state_dict()}}
        file_path = os.path.join(self.ckpt_dir, filename)
        with open(file_path, mode="wb+") as f:
            torch.save(states, f)
        if not silent:
            print("=> saved checkpoint '{}' (iter {})".format(file_path, self.global_iter))

    def load_checkpoint(self, filename):
        file_path = os.path.join(self.ckpt_dir, filename)
        if os.path.isfile(file_path):
            checkpoint = torch.load(file_path)
            self.global_iter = checkpoint["iter"]
            self.net.load_state_dict(checkpoint["model_states"]["net"])
            self.optim.load_state_dict(checkpoint["optim_states"]["optim"])
            print("=> loaded checkpoint '{} (iter {})'".format(file_path, self.global_iter))
        else:
            print("=> no checkpoint found at '{}'".format(file_path))

    def net_mode(self, train):
        if not isinstance(train, bool):
            raise ValueError("Only bool type is supported. True or False")
        if train:
            self.net.train()
        else:
            self.net.eval()
