"""Training driver for the MLP-mixing experiment: the counterpart of /root/reference/main_mlp.py on
the HIP hot path.  Same command-line flags (main_mlp.py:21-127), same phases (supervised then
unsupervised), same log lines and checkpoints (g.pth, sup_f.pth, unsup_f.pth).

    python -m cl_ica_amd.train_mlp --n 10 --space-type box --p 2 --batch-size 6144 --n-steps 1000
    python -m torch.distributed.run --nproc-per-node 8 -m cl_ica_amd.train_mlp ...      # data parallel

The unsupervised phase (the hot path) runs on ``ContrastiveTrainer``: on-device sampling, fused fp32-MFMA
encoder, tiled Lp-InfoNCE, fused Adam, replayed from a HIP graph on one GPU; with several ranks each GPU
trains on its own batch against the all-gathered negatives pool.
"""
from __future__ import annotations

import argparse
import os
import random

import numpy as np
import torch
import torch.nn.functional as F

from . import disentanglement_utils as du
from . import encoders, invertible_network_utils, latent_spaces, losses, spaces
from .distributed import gather_negatives, init_from_env
from .engine import ContrastiveTrainer, SamplerSpec
from .optim import Adam as FlatAdam

# (flag, type, default, help) -- the reference's CLI surface
_FLAGS = [
    ("--sphere-r", float, 1.0, None),
    ("--box-min", float, 0.0, "For box normalization only. Minimal value of box."),
    ("--box-max", float, 1.0, "For box normalization only. Maximal value of box."),
    ("--more-unsupervised", int, 3, "How many more steps to do for unsupervised compared to supervised training."),
    ("--save-dir", str, "", None),
    ("--num-eval-batches", int, 10, "Number of batches to average evaluation performance at the end."),
    ("--seed", int, None, None),
    ("--act-fct", str, "leaky_relu", "Activation function in mixing network g."),
    ("--c-param", float, 0.05, "Concentration parameter of the conditional distribution."),
    ("--m-param", float, 1.0, "Additional parameter for the marginal (only relevant if it is not uniform)."),
    ("--tau", float, 1.0, None),
    ("--n-mixing-layer", int, 3, "Number of layers in nonlinear mixing network g."),
    ("--n", int, 10, "Dimensionality of the latents."),
    ("--m-p", int, 0, "Type of ground-truth marginal distribution. p=0 means uniform; all other p values correspond to (projected) Lp Exponential"),
    ("--c-p", int, 2, "Exponent of ground-truth Lp Exponential distribution."),
    ("--lr", float, 1e-4, None),
    ("--p", int, 2, "Exponent of the assumed model Lp Exponential distribution."),
    ("--batch-size", int, 6144, None),
    ("--n-log-steps", int, 250, None),
    ("--n-steps", int, 100001, None),
]
_SWITCHES = [("--sphere-norm", "Normalize output to a sphere."), ("--box-norm", "Normalize output to a box."),
             ("--only-supervised", "Only train supervised model."), ("--only-unsupervised", "Only train unsupervised model."),
             ("--resume-training", None), ("--no-graph", "Eager launches instead of HIP-graph replay.")]


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="Disentanglement with InfoNCE/Contrastive Learning - MLP Mixing (MI355X)")
    for flag, typ, default, hlp in _FLAGS:
        ap.add_argument(flag, type=typ, default=default, help=hlp)
    for flag, hlp in _SWITCHES:
        ap.add_argument(flag, action="store_true", help=hlp)
    ap.add_argument("--space-type", type=str, default="box", choices=("box", "sphere", "unbounded"))
    args = ap.parse_args(argv)
    return args


_KIND = {0: None, 1: "laplace", 2: "normal"}


def sampler_spec(args, seed) -> SamplerSpec:
    space = {"box": "box", "sphere": "sphere", "unbounded": "real"}[args.space_type]
    marginal = "uniform" if args.m_p == 0 else _KIND.get(args.m_p, "gennorm")
    conditional = "vmf" if args.c_p == 0 else _KIND.get(args.c_p, "gennorm")
    return SamplerSpec(space=space, n=args.n, box=(args.box_min, args.box_max), marginal=marginal, m_param=args.m_param,
                       m_p=float(args.m_p), conditional=conditional, c_param=args.c_param, c_p=float(args.c_p), seed=seed)


def build_latent_space(args, spec: SamplerSpec):
    """LatentSpace with the reference's lambdas (main_mlp.py:136-194), on the device samplers."""
    if spec.space == "box":
        space = spaces.NBoxSpace(args.n, args.box_min, args.box_max)
    elif spec.space == "sphere":
        space = spaces.NSphereSpace(args.n, args.sphere_r)
    else:
        space = spaces.NRealSpace(args.n)
    eta = torch.zeros(args.n)
    if spec.space == "sphere":
        eta[0] = 1.0

    def marginal(sp, size, device="cuda"):
        if spec.marginal == "uniform":
            return sp.uniform(size, device=device)
        if spec.marginal == "gennorm":
            return sp.generalized_normal(eta, args.m_param, p=args.m_p, size=size, device=device)
        return getattr(sp, spec.marginal)(eta, args.m_param, size, device)

    def conditional(sp, z, size, device="cuda"):
        if spec.conditional == "vmf":
            return sp.von_mises_fisher(z, args.c_param, size, device)
        if spec.conditional == "gennorm":
            return sp.generalized_normal(z, args.c_param, p=args.c_p, size=size, device=device)
        return getattr(sp, spec.conditional)(z, args.c_param, size, device)

    return latent_spaces.LatentSpace(space=space, sample_marginal=marginal, sample_conditional=conditional)


def evaluate(h, latent_space, n_samples=4096):
    z = latent_space.sample_marginal(n_samples)
    with torch.no_grad():
        hz = h(z)
    (lin, _), _ = du.linear_disentanglement(z, hz, mode="r2")
    (perm, _), _ = du.permutation_disentanglement(z, hz, mode="pearson", solver="munkres", rescaling=True)
    return lin, perm


def autograd_train_step(h, loss, optimizer, z1, z2, supervised: bool, world: int = 1):
    """The reference's ``train_step`` verbatim in structure (main_mlp.py:258-285) on the drop-in modules, for the phases the
    fused engine does not cover: the SUPERVISED phase (``test = True``: ``F.mse_loss(z1_rec, z1)``, :274-276 -- the first of the
    default ``test_list = [True, False]``) and p = 0 (SimCLRLoss).  Returns the 0-dim loss tensor (no host sync)."""
    optimizer.zero_grad()
    z1_rec, z2_rec = h(z1), h(z2)
    # negatives = all z1_rec of the (global) batch: roll on one rank, autograd-aware all-gather on several
    z3_rec = torch.roll(z1_rec, 1, 0) if world == 1 else gather_negatives(z1_rec)
    if supervised:
        total = F.mse_loss(z1_rec, z1)
    else:
        total, _, _ = loss(z1, z2, torch.roll(z1, 1, 0), z1_rec, z2_rec, z3_rec)
    total.backward()
    optimizer.all_reduce_grads()
    optimizer.step()
    return total


def main(argv=None):
    return _main(argv)


def _main(argv=None):
    args = parse_args(argv)
    rank, world, device = init_from_env()
    if device.type != "cuda":
        raise SystemExit("cl_ica_amd.train_mlp needs an MI355X: there is no CPU path")
    log = print if rank == 0 else (lambda *a, **k: None)
    log("Arguments:")
    for k, v in vars(args).items():
        log(f"\t{k}: {v}")
    seed = args.seed if args.seed is not None else int.from_bytes(os.urandom(4), "little")
    if world > 1:      # without --seed every process would draw its own: rank 0's seed is THE seed (identical g and f everywhere)
        box = [seed]
        torch.distributed.broadcast_object_list(box, src=0)
        seed = int(box[0])
    np.random.seed(seed); random.seed(seed); torch.manual_seed(seed)   # same seeds on every rank: identical g and f
    spaces.manual_seed(seed * 1000003 + rank)
    spec = sampler_spec(args, seed)
    latent_space = build_latent_space(args, spec)
    loss = (losses.LpSimCLRLoss(p=args.p, tau=args.tau, simclr_compatibility_mode=True) if args.p
            else losses.SimCLRLoss(normalize=False, tau=args.tau))

    g = invertible_network_utils.construct_invertible_mlp(n=args.n, n_layers=args.n_mixing_layer, act_fct=args.act_fct,
                                                          cond_thresh_ratio=0.0, n_iter_cond_thresh=25000).to(device)
    lin, perm = evaluate(g, latent_space)
    log(f"Id. Lin. Disentanglement: {lin:.4f}")
    log(f"Id. Perm. Disentanglement: {perm:.4f}")
    if args.save_dir and rank == 0:
        os.makedirs(args.save_dir, exist_ok=True)
        torch.save(g.state_dict(), os.path.join(args.save_dir, "g.pth"))

    phases = [False] if args.only_unsupervised else ([True] if args.only_supervised else [True, False])
    total_loss_values = None
    f = None
    for supervised in phases:
        log(f"supervised test: {supervised}")
        if args.box_norm:
            out_norm = "learnable_box"
        elif args.sphere_norm:
            out_norm = "learnable_sphere"
        else:
            out_norm = "fixed_sphere" if args.p == 0 else None
        n = args.n
        f = encoders.get_mlp(n_in=n, n_out=n, layers=[n * 10, n * 50, n * 50, n * 50, n * 50, n * 10],
                             output_normalization=out_norm).to(device)
        log("f: ", f)
        h = lambda z: f(g(z))   # noqa: E731
        if total_loss_values is None or not args.resume_training:
            total_loss_values, lin_scores, perm_scores = [], [], []
        fused = (not supervised) and args.p != 0
        if fused:
            trainer = ContrastiveTrainer(f, g.weight_stack(), spec, batch_size=args.batch_size, p=args.p, tau=args.tau,
                                         lr=args.lr, g_slope=g.slope, g_act_kind=g.act_kind, device=device,
                                         process_group=None if world == 1 else torch.distributed.group.WORLD)
            if world == 1 and not args.no_graph:
                trainer.capture()
        else:
            # supervised / p == 0 phases: autograd over the drop-in modules; flat-arena HIP Adam whose gradient arena is
            # all-reduced under data parallelism (its 1/world average is applied inside the Adam launch)
            if world > 1:
                for prm in f.parameters():
                    torch.distributed.broadcast(prm.data, src=0)
            optimizer = FlatAdam(f.parameters(), lr=args.lr)

        def autograd_step():
            z1 = latent_space.sample_marginal(size=args.batch_size)
            z2 = latent_space.sample_conditional(z1, size=args.batch_size)
            return autograd_train_step(h, loss, optimizer, z1, z2, supervised, world)

        last_step = args.n_steps if supervised else args.n_steps * args.more_unsupervised
        global_step = len(total_loss_values) + 1
        pending = []     # device scalars, fetched only at log time: no host sync per step
        applied0, first_step = (trainer.steps_done if fused else 0), global_step
        while global_step <= last_step:
            pending.append(trainer.step()[0].clone() if fused else autograd_step().detach())
            if global_step % args.n_log_steps == 1 or global_step == args.n_steps:
                total_loss_values += [float(v) for v in torch.stack(pending).cpu()]
                pending = []
                lin, perm = evaluate(h, latent_space)
                log(f"Step: {global_step} \t", f"Loss: {total_loss_values[-1]:.4f} \t",
                    f"<Loss>: {np.mean(np.array(total_loss_values[-args.n_log_steps:])):.4f} \t",
                    f"Lin. Disentanglement: {lin:.4f} \t", f"Perm. Disentanglement: {perm:.4f}")
                if args.sphere_norm:
                    log(f"r: {f[-1].r}")
                if fused:
                    # f16x2 arithmetic: the device-side guard withholds a step whose tensors outgrew their scales and the next step redoes
                    # its batch (include/clica.h); here -- a log point, the host is synchronised anyway -- it is reported and checked
                    ga = trainer.check_arith()
                    if ga["new_skipped"]:
                        log(f"note: the f16x2 guard withheld {ga['new_skipped']} step(s) since the last log point (redone on fresh scales; "
                            f"{ga['skipped']} in total)")
                if fused and not getattr(trainer, "_guard_noted", False):
                    gs = trainer.loss_guard()          # (nothing to do: the library's device-side guard switches per step, include/clica.h)
                    if gs["limit"] > 0 and gs["last_spread"] > gs["limit"]:
                        trainer._guard_noted = True
                        log(f"note: the embeddings spread over M = {gs['last_spread']:.0f} temperature units (> {gs['limit']:.0f}): the p = 2 "
                            "loss runs on the coordinate-difference sweeps for such steps")
            lin_scores.append(lin); perm_scores.append(perm)
            global_step += 1
        if fused:
            # steps the guard withheld did not advance the device step counter: top the phase up to the number of APPLIED steps asked for
            want = last_step - first_step + 1
            for _ in range(64):
                if trainer.steps_done - applied0 >= want:
                    break
                pending.append(trainer.step()[0].clone())
        if pending:
            total_loss_values += [float(v) for v in torch.stack(pending).cpu()]
        if args.save_dir and rank == 0:
            torch.save(f.state_dict(), os.path.join(args.save_dir, "{}_f.pth".format("sup" if supervised else "unsup")))

    final_lin, final_perm = [], []
    h = lambda z: f(g(z))   # noqa: E731
    with torch.no_grad():
        for _ in range(args.num_eval_batches):
            z1 = latent_space.sample_marginal(size=args.batch_size)
            z1_rec = h(z1)
            (l, _), _ = du.linear_disentanglement(z1, z1_rec, mode="r2")
            (pm, _), _ = du.permutation_disentanglement(z1, z1_rec, mode="pearson", solver="munkres", rescaling=True)
            final_lin.append(l); final_perm.append(pm)
    engine_state = None
    if fused:      # what the engine ran on (f16x2 scales / overflow flag, the loss guard's counters): part of the run's record
        st, gs, ga = trainer.arith_state(), trainer.loss_guard(), trainer.check_arith()
        engine_state = dict(arith=st.get("arith"), f16_flags=st.get("flags"), f16_steps_withheld=ga["skipped"], loss_max_spread=gs["max_spread"],
                            loss_spread_limit=gs["limit"], loss_fallback_steps=gs["fallback_steps"])
        log(f"engine: encoder arithmetic {st.get('arith')}" + (f", scale flags {st.get('flags')}, steps withheld by the guard {ga['skipped']}" if "flags" in st else "") +
            f"; p = 2 loss guard: largest spread M = {gs['max_spread']:.0f} (limit {gs['limit']:.0f}), {gs['fallback_steps']} calls on the difference sweeps")
    log("linear mean: {} std: {}".format(np.mean(final_lin), np.std(final_lin)))
    log("perm mean: {} std: {}".format(np.mean(final_perm), np.std(final_perm)))
    if world > 1:
        torch.distributed.destroy_process_group()
    return dict(linear=float(np.mean(final_lin)), perm=float(np.mean(final_perm)), losses=total_loss_values, engine=engine_state)


if __name__ == "__main__":
    main()
