"""``train.py``-compatible driver for the MI355X engine.

Mirrors the command line, schedules, meters, checkpoint format and epoch loop of the reference's
``multimnist/train.py`` (flags ``:91-110``, ``AverageMeter`` ``:26-42``, ``save_checkpoint``/``load_checkpoint``
``:44-66``, LR schedule ``:132-137``, KL schedule ``:226-233``, ``train()`` ``:140-183``, ``test()`` ``:185-224``,
checkpoint dict ``:243-251``), with the batch loop body replaced by ONE fused enqueue (``FusedTrainer``).

    python -m multimodal_vae_amd.train --cuda --epochs 2 --synthetic 4096          # no data files needed
    python -m multimodal_vae_amd.train --cuda --data ./data                         # the reference's processed/*.pt

Differences that are deliberate: the loader is ``data.DeviceBatcher`` (uint8 H2D + on-device ToTensor, fixed batch
size: the ragged last batch of an epoch is dropped because a fused plan is built per batch size); loss values are read
back every ``--log_interval`` batches only (the reference synchronises three times per batch); sample dumps are written
only when ``--results`` is given (torchvision is not a dependency).
"""
from __future__ import annotations

import argparse
import os
import shutil
import sys

import torch

DEFAULT_N_LATENTS = 100


class AverageMeter(object):
    """multimnist/train.py:26-42"""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count


def save_checkpoint(state, is_best, folder='./', filename='checkpoint.pth.tar'):
    """multimnist/train.py:44-49 (same file names, same dict: see ``main``)."""
    os.makedirs(folder, exist_ok=True)
    torch.save(state, os.path.join(folder, filename))
    if is_best:
        shutil.copyfile(os.path.join(folder, filename), os.path.join(folder, 'model_best.pth.tar'))


def load_checkpoint(file_path, use_cuda=False):
    """multimnist/train.py:52-66: rebuilds a MultimodalVAE from a checkpoint written by either code base."""
    from .multimnist import MultimodalVAE
    checkpoint = torch.load(file_path, map_location=None if use_cuda else 'cpu', weights_only=False)
    n_latents = checkpoint['n_latents'] if 'n_latents' in checkpoint else DEFAULT_N_LATENTS
    vae = MultimodalVAE(n_latents=n_latents, use_cuda=use_cuda)
    vae.load_state_dict(checkpoint['state_dict'])
    if use_cuda:
        vae.cuda()
    return vae


def kl_schedule():
    """multimnist/train.py:227: the value advances every 5 epochs when --anneal_kl is given."""
    return iter([1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0])


def adjusted_lr(base_lr: float, epoch: int) -> float:
    """multimnist/train.py:132-137"""
    return base_lr * (0.1 ** (epoch // 5))


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    # the reference's flags, same names / defaults (multimnist/train.py:91-108)
    parser.add_argument('--n_latents', type=int, default=100, help='size of the latent embedding (default: 100)')
    parser.add_argument('--batch_size', type=int, default=128, metavar='N', help='input batch size for training (default: 128)')
    parser.add_argument('--epochs', type=int, default=20, metavar='N', help='number of epochs to train (default: 20)')
    parser.add_argument('--lr', type=float, default=1e-3, metavar='LR', help='learning rate (default: 1e-3)')
    parser.add_argument('--log_interval', type=int, default=10, metavar='N', help='how many batches to wait before logging training status')
    parser.add_argument('--anneal_kl', action='store_true', default=False, help='if True, use a fixed interval of doubling the KL term')
    parser.add_argument('--anneal_lr', action='store_true', default=False, help='If True, half learning rate every 5 epochs')
    parser.add_argument('--cuda', action='store_true', default=False, help='enables CUDA training')
    # additions
    parser.add_argument('--data', type=str, default='./data', help="root of the reference's processed/{training,test}.pt")
    parser.add_argument('--synthetic', type=int, default=0, metavar='N', help='train on N synthetic MultiMNIST-shaped samples instead of files')
    parser.add_argument('--out', type=str, default='./trained_models', help='checkpoint folder (reference: ./trained_models)')
    parser.add_argument('--results', type=str, default='', help='folder for per-epoch sample dumps (off when empty)')
    parser.add_argument('--seed', type=int, default=1234)
    return parser


def main(argv=None) -> dict:
    """Single GPU: ``python -m multimodal_vae_amd.train --cuda ...``.  Data parallel (BASELINE configuration 4, one process
    per GPU): ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 -m
    multimodal_vae_amd.train --cuda ...`` -- every rank trains on its own shard of the training set with the per-GPU
    ``--batch_size``, gradients are summed over RCCL (1/world inside Adam), rank 0 prints and writes checkpoints, the
    BatchNorm running statistics are averaged over the ranks before every test pass / checkpoint."""
    from . import dp
    dp.ensure_ipc_env()                  # before the first GPU call (torch.cuda.is_available() below is one)
    args = build_parser().parse_args(argv)
    args.cuda = args.cuda and torch.cuda.is_available()
    if not args.cuda:
        raise SystemExit("this engine runs on a gfx950 GPU only: pass --cuda on a machine that has one (no CPU fallback)")
    from . import data as D
    from .multimnist import FusedTrainer, MultimodalVAE
    from .utils import tensor_to_string

    world, local = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        torch.cuda.set_device(local)
    dev = torch.device("cuda", torch.cuda.current_device())
    rank, world, _ = dp.init_distributed("nccl", dev)
    chief = rank == 0
    if not chief:
        global print
        print = lambda *a, **k: None     # noqa: E731  (rank 0 reports)
    torch.manual_seed(args.seed)
    if args.synthetic > 0:
        n_test = max(args.batch_size, args.synthetic // 6)
        tr_x, tr_y = D.synthetic_multimnist(args.synthetic, seed=args.seed)
        te_x, te_y = D.synthetic_multimnist(n_test, seed=args.seed + 1)
        from .utils import charlist_tensor
        tr_t = torch.stack([charlist_tensor(l) for l in tr_y]); te_t = torch.stack([charlist_tensor(l) for l in te_y])
    else:
        tr_x, tr_t = D.load_multimnist(args.data, train=True)
        te_x, te_t = D.load_multimnist(args.data, train=False)
    if world > 1:                        # rank r trains / tests on samples r, r + world, ... (equal shard sizes)
        n_tr, n_te = len(tr_x) // world * world, len(te_x) // world * world
        tr_x, tr_t, te_x, te_t = (t.contiguous() for t in (tr_x[rank:n_tr:world], tr_t[rank:n_tr:world],
                                                           te_x[rank:n_te:world], te_t[rank:n_te:world]))
    train_loader = D.DeviceBatcher(tr_x, tr_t, args.batch_size, dev, shuffle=True, seed=dp.rank_seed(args.seed, rank))
    test_loader = D.DeviceBatcher(te_x, te_t, args.batch_size, dev, shuffle=True, seed=dp.rank_seed(args.seed + 7, rank))

    if len(train_loader) == 0 or len(test_loader) == 0:
        import warnings
        warnings.warn("rank %d: %d training / %d test samples give no full batch of %d: nothing to do in that phase"
                      % (rank, len(tr_x), len(te_x), args.batch_size))

    vae = MultimodalVAE(args.n_latents, use_cuda=True).cuda()
    trainer = FusedTrainer(vae, args.batch_size, lr=args.lr, kl_lambda=1e-3, seed=dp.rank_seed(args.seed, rank), world_size=world,
                           all_reduce=dp.GradAllReduce() if world > 1 else None)
    state = trainer.engine.state
    dp.broadcast_flat(state.params)      # identical replicas (every rank seeds the same initialisation anyway)
    dp.broadcast_flat(state.bn_stats)
    if world > 1:
        state.pack_weights()

    def train(epoch, kl_lambda):
        vae.train()
        joint_loss_meter, image_loss_meter, text_loss_meter = AverageMeter(), AverageMeter(), AverageMeter()
        n_total = len(train_loader) * args.batch_size
        pending = []
        for batch_idx, (image, text) in enumerate(train_loader):
            out = trainer(image, text, kl_lambda=kl_lambda)
            pending.append(out.losses().clone())                    # device tensor, no sync
            if batch_idx % args.log_interval == 0:
                for l in torch.stack(pending).cpu().tolist():        # one read-back per log interval
                    joint_loss_meter.update(l[0], args.batch_size)
                    image_loss_meter.update(l[1], args.batch_size)
                    text_loss_meter.update(l[2], args.batch_size)
                pending = []
                print('Train Epoch: {} [{}/{} ({:.0f}%)]\tJoint Loss: {:.6f}\tImage Loss: {:.6f}\tText Loss: {:.6f}'.format(
                    epoch, batch_idx * args.batch_size, n_total, 100. * batch_idx / max(len(train_loader), 1),
                    joint_loss_meter.avg, image_loss_meter.avg, text_loss_meter.avg))
        if pending:
            for l in torch.stack(pending).cpu().tolist():
                joint_loss_meter.update(l[0], args.batch_size)
                image_loss_meter.update(l[1], args.batch_size)
                text_loss_meter.update(l[2], args.batch_size)
        print('====> Epoch: {}\tJoint loss: {:.4f}\tImage loss: {:.4f}\tText loss: {:.4f}'.format(
            epoch, joint_loss_meter.avg, image_loss_meter.avg, text_loss_meter.avg))
        return joint_loss_meter.avg, image_loss_meter.avg, text_loss_meter.avg

    def test(kl_lambda):
        vae.eval()
        trainer.engine.kl_lambda = kl_lambda
        dp.sync_bn_buffers(state.bn_stats, state.bn_nbt)     # eval (and the checkpoint below) see the ranks' average
        acc = torch.zeros(3, device=dev)
        nb = 0
        for image, text in test_loader:
            acc += trainer.evaluate(image, text).losses()
            nb += 1
        acc = acc / max(nb, 1)
        if world > 1:
            torch.distributed.all_reduce(acc)
            acc = acc / world
        j, i, t = acc.cpu().tolist()
        print('====> Test Epoch\tJoint loss: {:.4f}\tImage loss: {:.4f}\tText loss:{:.4f}'.format(j, i, t))
        return j + i + t, (j, i, t)

    # everything allocated so far lives for the whole run: keep the cyclic collector from re-scanning it in the enqueue
    # thread (a generational collection there stalls the GPU for milliseconds; bench.py: 0.959 -> 0.940 ms per step)
    import gc
    gc.collect()
    gc.freeze()
    kl_lambda = 1e-3
    schedule = kl_schedule()
    best_loss = float(sys.maxsize)
    history = {"train": [], "test": []}
    for epoch in range(1, args.epochs + 1):
        if (epoch - 1) % 5 == 0 and args.anneal_kl:
            kl_lambda = next(schedule, kl_lambda)
        history["train"].append(train(epoch, kl_lambda))
        loss, (joint_loss, image_loss, text_loss) = test(kl_lambda)
        history["test"].append((joint_loss, image_loss, text_loss))
        if args.anneal_lr:
            trainer.engine.lr = adjusted_lr(args.lr, epoch)
            print('learning rate: {:.4f}'.format(trainer.engine.lr))
        is_best = loss < best_loss
        best_loss = min(loss, best_loss)
        eng = trainer.engine
        if chief:
            save_checkpoint({
                'state_dict': vae.state_dict(),
                'best_loss': best_loss,
                'joint_loss': joint_loss,
                'image_loss': image_loss,
                'text_loss': text_loss,
                'n_latents': args.n_latents,
                # torch.optim.Adam.state_dict() layout, so that optim.Adam(...).load_state_dict() accepts it
                'optimizer': adam_state_dict(vae, eng),
            }, is_best, folder=args.out)
        if args.results and chief:
            os.makedirs(args.results, exist_ok=True)
            sample = torch.randn(64, args.n_latents, device=dev)
            vae.eval()
            with torch.no_grad():
                image_sample = vae.image_decoder(sample).cpu()
                words = vae.text_decoder(sample)                       # (64, 4, 12) log-probs
            torch.save(image_sample.view(64, 1, 50, 50), os.path.join(args.results, 'sample_image_epoch%d.pt' % epoch))
            # the reference samples torch.multinomial from LOG-probabilities (multimnist/model.py:290-296), which modern
            # torch rejects; the dump uses the greedy path the decoder itself feeds back
            text_sample = words.argmax(dim=2).cpu()
            with open(os.path.join(args.results, 'sample_text_epoch%d.txt' % epoch), 'w') as fp:
                for i in range(text_sample.size(0)):
                    fp.write('%s\n' % tensor_to_string(text_sample[i]))
    if world > 1:
        dp.barrier(dev)
        torch.distributed.destroy_process_group()
    return history


def adam_state_dict(vae, engine) -> dict:
    """The fused engine's flat Adam moments as a ``torch.optim.Adam.state_dict()`` (per-parameter views, cloned)."""
    st = engine.state
    step = int(engine.adam_state[0].item())
    state = {}
    for i, (name, shape, off) in enumerate(st.table):
        numel = 1
        for s in shape:
            numel *= s
        state[i] = {'step': torch.tensor(float(step)),
                    'exp_avg': engine.exp_avg[off:off + numel].view(shape).clone(),
                    'exp_avg_sq': engine.exp_avg_sq[off:off + numel].view(shape).clone()}
    group = {'lr': engine.lr, 'betas': tuple(engine.betas), 'eps': engine.eps, 'weight_decay': 0, 'amsgrad': False,
             'maximize': False, 'foreach': None, 'capturable': False, 'differentiable': False, 'fused': None,
             'decoupled_weight_decay': False, 'params': list(range(len(st.table)))}
    return {'state': state, 'param_groups': [group]}


if __name__ == "__main__":
    main()
