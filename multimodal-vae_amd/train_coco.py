"""``coco/train.py``-compatible driver for the MI355X engine (SURVEY §8d configuration 5).

Mirrors the command line (``coco/train.py:88-104``: ``--n_latents 100 --batch_size 64 --epochs 20 --lr 1e-4
--log_interval 10 --anneal_kl --cuda``), the KL schedule (``:223-231``), the epoch loop and the checkpoint dict
(``:233-248``) of the reference, with the batch loop body (``:138-173``) replaced by ONE fused enqueue and the
``DataLoader`` by ``data.DeviceBatcher``: uint8 pixels and fp32 caption vectors are gathered into pinned staging buffers
and copied one batch ahead on a copy stream; ToTensor runs on the device.

    python -m multimodal_vae_amd.train_coco --cuda --epochs 2 --synthetic 4096      # no data files needed

Inputs.  The reference builds its batches from the COCO caption files + the 2 GB GloVe table (torchvision / torchtext /
nltk, none of which this engine depends on).  This driver takes tensors: ``--data DIR`` with ``images_u8.pt``
((N,3,32,32) uint8: Scale(32) + CenterCrop(32), coco/train.py:107-112), ``captions.pt`` ((N,102,300) fp32 from
``text_transformer``, coco/utils.py:18-49) and ``sos.pt`` (GloVe('<s>'), 300 floats), or ``--synthetic N``.
"""
from __future__ import annotations

import argparse
import os
import sys

import torch

from .train import AverageMeter, adam_state_dict, kl_schedule, save_checkpoint

MAX_WORDS = 102


def build_parser() -> argparse.ArgumentParser:
    parser = argparse.ArgumentParser()
    # the reference's flags, same names / defaults (coco/train.py:88-104)
    parser.add_argument('--n_latents', type=int, default=100, help='size of the latent embedding')
    parser.add_argument('--batch_size', type=int, default=64, metavar='N', help='input batch size for training (default: 64)')
    parser.add_argument('--epochs', type=int, default=20, metavar='N', help='number of epochs to train (default: 20)')
    parser.add_argument('--lr', type=float, default=1e-4, metavar='LR', help='learning rate (default: 1e-4)')
    parser.add_argument('--log_interval', type=int, default=10, metavar='N', help='how many batches to wait before logging training status (default: 10)')
    parser.add_argument('--anneal_kl', action='store_true', default=False, help='if True, use a fixed interval of doubling the KL term')
    parser.add_argument('--cuda', action='store_true', default=False, help='enables CUDA training')
    # additions
    parser.add_argument('--data', type=str, default='./data/coco', help='folder with images_u8.pt, captions.pt, sos.pt')
    parser.add_argument('--synthetic', type=int, default=0, metavar='N', help='train on N synthetic COCO-shaped samples instead of files')
    parser.add_argument('--out', type=str, default='./trained_models', help='checkpoint folder (reference: ./trained_models)')
    parser.add_argument('--results', type=str, default='', help='folder for per-epoch sample dumps (off when empty)')
    parser.add_argument('--seed', type=int, default=1234)
    return parser


def synthetic_coco(n: int, seed: int = 0):
    """COCO-shaped stand-ins: uint8 colour images, caption tensors with GloVe-like rows for the first 8..47 positions
    and zero rows after (coco/utils.py:40-47), and a '<s>' vector."""
    g = torch.Generator().manual_seed(seed)
    images = torch.randint(0, 256, (n, 3, 32, 32), dtype=torch.uint8, generator=g)
    captions = 0.4 * torch.randn(n, MAX_WORDS, 300, generator=g)
    lens = torch.randint(8, 48, (n,), generator=g)
    captions *= (torch.arange(MAX_WORDS).view(1, -1, 1) < lens.view(-1, 1, 1)).float()
    sos = 0.4 * torch.randn(300, generator=torch.Generator().manual_seed(999))
    return images, captions, sos


def main(argv=None) -> dict:
    args = build_parser().parse_args(argv)
    args.cuda = args.cuda and torch.cuda.is_available()
    if not args.cuda:
        raise SystemExit("this engine runs on a gfx950 GPU only: pass --cuda on a machine that has one (no CPU fallback)")
    from . import data as D
    from .coco import FusedTrainer, MultimodalVAE

    dev = torch.device("cuda", torch.cuda.current_device())
    torch.manual_seed(args.seed)
    if args.synthetic > 0:
        n_test = max(args.batch_size, args.synthetic // 6)
        tr_x, tr_t, sos = synthetic_coco(args.synthetic, seed=args.seed)
        te_x, te_t, _ = synthetic_coco(n_test, seed=args.seed + 1)
    else:
        tr_x = torch.load(os.path.join(args.data, "images_u8.pt"))
        tr_t = torch.load(os.path.join(args.data, "captions.pt"))
        sos = torch.load(os.path.join(args.data, "sos.pt"))
        n_test = max(args.batch_size, len(tr_x) // 10)
        te_x, te_t, tr_x, tr_t = tr_x[:n_test], tr_t[:n_test], tr_x[n_test:], tr_t[n_test:]
    train_loader = D.DeviceBatcher(tr_x, tr_t, args.batch_size, dev, shuffle=True, seed=args.seed)
    test_loader = D.DeviceBatcher(te_x, te_t, args.batch_size, dev, shuffle=True, seed=args.seed + 7)

    vae = MultimodalVAE(args.n_latents, use_cuda=True, sos=sos).cuda()
    trainer = FusedTrainer(vae, args.batch_size, lr=args.lr, kl_lambda=1e-3, seed=args.seed)

    def train(epoch, kl_lambda):
        vae.train()
        trainer.engine.kl_lambda = kl_lambda
        meters = AverageMeter(), AverageMeter(), AverageMeter()
        n_total = len(train_loader) * args.batch_size
        pending = []

        def drain():
            for l in torch.stack(pending).cpu().tolist():        # one read-back per log interval
                for m, v in zip(meters, l):
                    m.update(v, args.batch_size)
            pending.clear()

        for batch_idx, (image, text) in enumerate(train_loader):
            pending.append(trainer(image, text).losses().clone())   # device tensor, no sync
            if batch_idx % args.log_interval == 0:
                drain()
                print('Train Epoch: {} [{}/{} ({:.0f}%)]\tJoint Loss: {:.6f}\tImage Loss: {:.6f}\tText Loss: {:.6f}'.format(
                    epoch, batch_idx * args.batch_size, n_total, 100. * batch_idx / max(len(train_loader), 1),
                    meters[0].avg, meters[1].avg, meters[2].avg))
        if pending:
            drain()
        print('====> Epoch: {}\tJoint loss: {:.4f}\tImage loss: {:.4f}\tText loss: {:.4f}'.format(
            epoch, meters[0].avg, meters[1].avg, meters[2].avg))
        return meters[0].avg, meters[1].avg, meters[2].avg

    def test(kl_lambda):
        vae.eval()
        trainer.engine.kl_lambda = kl_lambda
        acc = torch.zeros(3, device=dev)
        nb = 0
        for image, text in test_loader:
            acc += trainer.evaluate(image, text).losses()
            nb += 1
        j, i, t = (acc / max(nb, 1)).cpu().tolist()
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
        is_best = loss < best_loss
        best_loss = min(loss, best_loss)
        save_checkpoint({
            'state_dict': vae.state_dict(),
            'best_loss': best_loss,
            'joint_loss': joint_loss,
            'image_loss': image_loss,
            'text_loss': text_loss,
            'n_latents': args.n_latents,
            'optimizer': adam_state_dict(vae, trainer.engine),
        }, is_best, folder=args.out)
        if args.results:
            os.makedirs(args.results, exist_ok=True)
            sample = torch.randn(64, args.n_latents, device=dev)
            vae.eval()
            with torch.no_grad():
                torch.save(vae.image_decoder(sample).cpu().view(64, 3, 32, 32), os.path.join(args.results, 'sample_image_epoch%d.pt' % epoch))
                torch.save(vae.text_decoder.generate_vector(sample).cpu(), os.path.join(args.results, 'sample_text_vector.pt'))
    return history


if __name__ == "__main__":
    main()
