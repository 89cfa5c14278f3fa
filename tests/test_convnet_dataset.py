"""Config-5 pieces: the ConvNet restatement (shape / parameter-count / padding known answers; TensorFlow is not
available to produce golden outputs) and the dataset wire format."""
import numpy as np
import torch

from softgrip_amd import convnet as cn
from softgrip_amd import dataset as ds


def test_param_counts_match_keras_summary():
    m = cn.ConvNet()
    cnt = lambda mod: sum(p.numel() for p in mod.parameters())
    assert cnt(m.conv1) == 4736 and cnt(m.conv2) == 98560 and cnt(m.conv3) == 393728
    assert cnt(m.fc1) == 262656 and cnt(m.fc2) == 131328 and cnt(m.fc3) == 32896 and cnt(m.fc4) == 8256 and cnt(m.out) == 65
    trainable = sum(p.numel() for p in m.parameters())
    assert trainable == 934785
    moving = sum(b.numel() for n, b in m.named_buffers() if "running" in n)
    assert moving == 2560


def test_shapes_and_same_padding():
    m = cn.ConvNet().eval()
    x = torch.randn(5, 200, 12, dtype=torch.float64)
    assert m(x).shape == (5, 1)
    # TF SAME, k=3, s=2, even length: output t sees inputs 2t, 2t+1, 2t+2 with a zero appended on the right only
    conv = torch.nn.Conv1d(1, 1, 3, stride=2, bias=False)
    with torch.no_grad():
        conv.weight[:] = torch.tensor([[[1.0, 10.0, 100.0]]])
    sig = torch.arange(1.0, 9.0).reshape(1, 1, 8)
    out = conv(cn.ConvNet._same(sig)).flatten().tolist()
    assert out == [1 + 20 + 300, 3 + 40 + 500, 5 + 60 + 700, 7 + 80 + 0]
    for L, Lout in ((200, 100), (100, 50), (50, 25)):
        assert conv(cn.ConvNet._same(torch.zeros(1, 1, L))).shape[-1] == Lout
    p = cn.normalize_predictions(torch.tensor([[-1e9], [0.0], [1e9]]))
    assert p.tolist() == [300.0, 850.0, 1400.0]


def test_train_step_decreases_loss():
    torch.manual_seed(0)
    m = cn.ConvNet()
    opt = cn.make_optimizer(m)          # training_cross_validate.py:58-61 (constant 1e-3)
    x = torch.randn(32, 200, 12, dtype=torch.float64)
    y = 300 + 1100 * torch.rand(32)
    mean, std = cn.channel_stats(x)
    assert mean.shape == (1, 1, 12) and std.shape == (1, 1, 12)
    losses = [float(cn.train_step(m, opt, x, y, mean, std)[0]) for _ in range(12)]
    assert losses[-1] < losses[0]
    n = cn.noised_modality(torch.zeros(2000, 10, 12, dtype=torch.float64))
    assert abs(float(n[..., :6].std()) - 0.7) < 0.02 and abs(float(n[..., 6:].std()) - 0.06) < 0.003


def test_dataset_roundtrip_and_stats(tmp_path):
    rng = np.random.RandomState(0)
    data = [rng.randn(200, 12) for _ in range(6)]
    k = list(rng.uniform(300, 1400, 6))
    p1, p2 = tmp_path / "a.pickle", tmp_path / "b.pickle"
    ds.save_dataset(p1, data[:4], k[:4])
    ds.save_dataset(p2, data[4:], k[4:])
    d = ds.load_datasets([p1, p2])
    assert len(d["data"]) == 6 and d["stiffness"] == [float(x) for x in k]
    assert d["data"][0].dtype == np.float64 and d["data"][0].shape == (200, 12)
    tx, ty, vx, vy, mean, std = ds.split_and_stats(d, [0, 1, 2, 3], [4, 5])
    assert tx.shape == (4, 200, 12) and vx.shape == (2, 200, 12) and mean.shape == (1, 1, 12)
    np.testing.assert_allclose(mean[0, 0], np.concatenate(data[:4]).mean(0), atol=1e-12)
    np.testing.assert_allclose(std[0, 0], np.concatenate(data[:4]).std(0), atol=1e-12)


def test_conv1_bn_relu_hand_computed_train_and_eval():
    """Pin of the first block (net/layers.py:24-29: Conv1D(k 3, s 2, SAME) -> BatchNormalization -> ReLU) against a computation by
    hand in NumPy -- Keras outputs cannot be captured here (TensorFlow absent): 2 samples, 8 steps, 12 channels, float64.
    Training mode: batch statistics over (batch, time) with the population variance; the moving statistics take one 0.99 / 0.01 step.
    Inference mode: the moving statistics.  TF SAME on an even length pads one zero on the right."""
    torch.manual_seed(3)
    rng = np.random.RandomState(3)
    net = cn.ConvNet().double()
    x = rng.randn(2, 8, 12)
    W = net.conv1.weight.detach().numpy()        # [128, 12, 3]
    b = rng.randn(128) * 0.1
    with torch.no_grad():
        net.conv1.bias[:] = torch.tensor(b)
    xp = np.concatenate([x, np.zeros((2, 1, 12))], axis=1)                       # SAME: (0, 1)
    conv = np.zeros((2, 4, 128))
    for n in range(2):
        for t in range(4):
            win = xp[n, 2 * t:2 * t + 3]                                          # [3, 12]: inputs 2t, 2t+1, 2t+2
            conv[n, t] = np.einsum("kc,ock->o", win, W) + b
    mean, var = conv.mean(axis=(0, 1)), conv.var(axis=(0, 1))                    # population variance
    want_train = np.maximum((conv - mean) / np.sqrt(var + 1e-3), 0.0)

    def block(xx):
        h = net.conv1(cn.ConvNet._same(torch.tensor(xx).transpose(1, 2)))
        return torch.relu(net.bn1(h)).transpose(1, 2).detach().numpy()
    net.train()
    np.testing.assert_allclose(block(x), want_train, atol=1e-12)
    np.testing.assert_allclose(net.bn1.running_mean.numpy(), 0.01 * mean, atol=1e-14)
    np.testing.assert_allclose(net.bn1.running_var.numpy(), 0.99 + 0.01 * var, atol=1e-14)
    net.eval()
    want_eval = np.maximum((conv - 0.01 * mean) / np.sqrt(0.99 + 0.01 * var + 1e-3), 0.0)
    np.testing.assert_allclose(block(x), want_eval, atol=1e-12)
    # dense block (layers.py:43-47): batch statistics over the batch axis only
    bn = cn.KerasBatchNorm(5).double().train()
    z = rng.randn(7, 5)
    np.testing.assert_allclose(bn(torch.tensor(z)).detach().numpy(), (z - z.mean(0)) / np.sqrt(z.var(0) + 1e-3), atol=1e-12)
    np.testing.assert_allclose(bn.running_var.numpy(), 0.99 + 0.01 * z.var(0), atol=1e-14)


def _state(net):
    return {k: v.detach().cpu().double().numpy().copy() for k, v in net.state_dict().items()}


def _random_net(seed, dtype=torch.float64):
    """a ConvNet with every parameter and moving statistic away from its initial value (biases, gammas, betas, moving mean / variance)"""
    torch.manual_seed(seed)
    net = cn.ConvNet().to(dtype)
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for name, p in net.named_parameters():
            if name.endswith("bias"):
                p.copy_(0.1 * torch.randn(p.shape, generator=g, dtype=dtype))
            elif "bn" in name and name.endswith("weight"):
                p.copy_(1.0 + 0.2 * torch.randn(p.shape, generator=g, dtype=dtype))
        for name, b in net.named_buffers():
            b.copy_((0.1 * torch.randn(b.shape, generator=g, dtype=dtype)) if name.endswith("mean") else (0.5 + torch.rand(b.shape, generator=g, dtype=dtype)))
    return net


def test_whole_network_against_the_numpy_restatement_train_and_eval():
    """The hand-computed pin extended from conv1 + BN to the WHOLE network: conv2 / conv3 (no BN / activation on the
    last conv), global average pooling, the three dense + BN + ReLU blocks, dense 64, dense 1, the head and the MAE loss -- module
    (fp64) against tests/np_convnet.py, which shares no code with it, in training mode (batch statistics, moving statistics updated by
    one 0.99 / 0.01 step) and in inference mode (moving statistics), on an even and on an odd number of time steps."""
    import np_convnet as npc
    rng = np.random.RandomState(5)
    for T in (200, 26, 9):
        net = _random_net(11 + T)
        x = rng.randn(6, T, 12)
        y = rng.uniform(300, 1400, 6)
        p = _state(net)
        net.eval()
        with torch.no_grad():
            got_eval = net(torch.tensor(x))[:, 0].numpy()
        np.testing.assert_allclose(got_eval, npc.forward(p, x, training=False), rtol=1e-10, atol=1e-10)
        net.train()
        new_stats = {}
        want_train = npc.forward(p, x, training=True, new_stats=new_stats)
        with torch.no_grad():
            got_train = net(torch.tensor(x))[:, 0].numpy()
        np.testing.assert_allclose(got_train, want_train, rtol=1e-9, atol=1e-9)
        after = _state(net)
        assert len(new_stats) == 10
        for k, v in new_stats.items():
            np.testing.assert_allclose(after[k], v, rtol=1e-12, atol=1e-12)
        pred = cn.normalize_predictions(torch.tensor(want_train)[:, None]).numpy()
        np.testing.assert_allclose(pred, npc.predictions(want_train), rtol=1e-12)
        assert abs(float(np.abs(pred - y).mean()) - npc.mae_loss(p, x, y)) < 1e-9


def test_gradients_against_finite_differences_of_the_numpy_loss_and_adam_step():
    """every gradient tensor of one training step (autograd, fp64) against central finite differences of the NumPy restatement's loss
    along a random direction per tensor, and the optimizer step against Adam's first step written out by hand with Keras' epsilon"""
    import np_convnet as npc
    rng = np.random.RandomState(7)
    net = _random_net(3)
    x = rng.randn(5, 24, 12)
    y = rng.uniform(300, 1400, 5)
    one, zero = torch.ones(1, 1, 12, dtype=torch.float64), torch.zeros(1, 1, 12, dtype=torch.float64)
    p0 = _state(net)
    opt = cn.make_optimizer(net)
    loss, _ = cn.train_step(net, opt, torch.tensor(x), torch.tensor(y), zero, one)
    assert abs(float(loss) - npc.mae_loss(p0, x, y)) < 1e-9
    grads = {k: v.grad.numpy().copy() for k, v in net.named_parameters()}
    assert len(grads) == 26
    for k, g in grads.items():
        d = rng.randn(*g.shape)
        d /= np.linalg.norm(d)
        eps = 1e-6
        pp, pm = dict(p0), dict(p0)
        pp[k], pm[k] = p0[k] + eps * d, p0[k] - eps * d
        fd = (npc.mae_loss(pp, x, y) - npc.mae_loss(pm, x, y)) / (2 * eps)
        an = float((g * d).sum())
        assert abs(fd - an) <= 1e-5 * max(1.0, abs(an)), (k, fd, an)
    after = _state(net)
    for k, g in grads.items():
        np.testing.assert_allclose(after[k], npc.adam_first_step(p0[k], g), rtol=0, atol=1e-12, err_msg=k)
