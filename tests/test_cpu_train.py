"""CPU checks of the train.py-compatible driver and the input-format helpers."""
import os

import pytest
import torch


def test_reference_cli_flags_and_schedules():
    from multimodal_vae_amd import train as T
    p = T.build_parser()
    a = p.parse_args([])
    # multimnist/train.py:91-108: names and defaults
    assert (a.n_latents, a.batch_size, a.epochs, a.lr, a.log_interval) == (100, 128, 20, 1e-3, 10)
    assert a.anneal_kl is False and a.anneal_lr is False and a.cuda is False
    b = p.parse_args("--n_latents 20 --batch_size 64 --epochs 3 --lr 0.01 --log_interval 5 --anneal_kl --anneal_lr --cuda".split())
    assert (b.n_latents, b.batch_size, b.epochs, b.lr, b.log_interval, b.anneal_kl, b.anneal_lr, b.cuda) == (20, 64, 3, 0.01, 5, True, True, True)
    assert list(T.kl_schedule()) == [1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 1.0]                    # multimnist/train.py:227
    assert [T.adjusted_lr(1e-3, e) for e in (1, 4, 5, 9, 10)] == [1e-3, 1e-3, 1e-4, 1e-4, pytest.approx(1e-5)]   # :132-137
    m = T.AverageMeter()
    m.update(2.0, 4); m.update(4.0, 4)
    assert (m.val, m.sum, m.count, m.avg) == (4.0, 24.0, 8, 3.0)
    with pytest.raises(SystemExit):
        T.main(["--epochs", "1"])                                                        # no --cuda: refuses (no CPU fallback)


def test_checkpoint_format_round_trip(tmp_path):
    from multimodal_vae_amd import train as T
    from multimodal_vae_amd.multimnist import MultimodalVAE
    vae = MultimodalVAE(20)
    sd = {k: v.clone() for k, v in vae.state_dict().items()}
    T.save_checkpoint({'state_dict': vae.state_dict(), 'best_loss': 1.0, 'joint_loss': 0.5, 'image_loss': 0.3, 'text_loss': 0.2,
                       'n_latents': 20, 'optimizer': {}}, True, folder=str(tmp_path))
    assert os.path.exists(tmp_path / 'checkpoint.pth.tar') and os.path.exists(tmp_path / 'model_best.pth.tar')   # train.py:44-49
    back = T.load_checkpoint(str(tmp_path / 'model_best.pth.tar'), use_cuda=False)
    assert back.n_latents == 20
    for k, v in back.state_dict().items():
        assert torch.equal(v, sd[k]), k


def test_text_utils_match_reference_semantics():
    from multimodal_vae_amd import utils as U
    assert (U.max_length, U.n_characters, U.SOS, U.FILL) == (4, 12, 10, 11)
    assert U.char_tensor("37").tolist() == [3, 7, 11, 11]
    assert U.charlist_tensor([4, 0, 9]).tolist() == [4, 0, 9, 11] and U.charlist_tensor([]).tolist() == [11] * 4
    assert U.tensor_to_string(torch.tensor([10, 5, 11, 2])) == "^52"                     # SOS -> '^', FILL -> ''


def test_multimnist_file_format_round_trip(tmp_path):
    from multimodal_vae_amd import data as D
    x, y = D.synthetic_multimnist(37, seed=3)
    assert x.shape == (37, 50, 50) and x.dtype == torch.uint8 and len(y) == 37 and all(len(l) <= 4 for l in y)
    path = D.save_multimnist(str(tmp_path), True, x, y)
    assert path.endswith(os.path.join("processed", "training.pt"))                        # multimnist/datasets.py:26-30
    raw = torch.load(path, weights_only=False)
    assert isinstance(raw, tuple) and raw[0].dtype == torch.uint8 and isinstance(raw[1], list)   # :180-190
    xi, ti = D.load_multimnist(str(tmp_path), True)
    assert torch.equal(xi, x) and ti.shape == (37, 4)
    for l, t in zip(y, ti.tolist()):
        assert t == l + [11] * (4 - len(l))
    with pytest.raises(RuntimeError):
        D.load_multimnist(str(tmp_path), False)


def test_coco_driver_flags_and_synthetic_inputs():
    from multimodal_vae_amd import train_coco as T
    a = T.build_parser().parse_args([])
    # coco/train.py:88-104: names and defaults
    assert (a.n_latents, a.batch_size, a.epochs, a.lr, a.log_interval, a.anneal_kl, a.cuda) == (100, 64, 20, 1e-4, 10, False, False)
    x, t, sos = T.synthetic_coco(9, seed=1)
    assert x.shape == (9, 3, 32, 32) and x.dtype == torch.uint8 and t.shape == (9, 102, 300) and sos.shape == (300,)
    assert (t[:, -1] == 0).all() and (t[:, 0] != 0).any()                # zero rows after the caption (coco/utils.py:40-47)
    with pytest.raises(SystemExit):
        T.main(["--epochs", "1"])                                        # no --cuda: refuses (no CPU fallback)
