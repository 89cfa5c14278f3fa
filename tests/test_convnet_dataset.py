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
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)          # training_cross_validate.py:58-61 (constant 1e-3)
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
