"""NumPy fp64 restatement of the reference's ConvNet forward pass and of one Adam step -- TEST INFRASTRUCTURE: an evaluation that shares
no code with soft-grip_amd/convnet.py (no torch), written from the reference's layer list (net/NeuralNets.py:6-27, net/layers.py:13-52)
and Keras' documented semantics: Conv1D(k 3, s 2, "SAME") = one zero appended on the right of an even-length signal;
BatchNormalization(momentum 0.99, epsilon 1e-3) = batch mean / POPULATION variance over every axis but the channel in training, the moving
statistics at inference; GlobalAveragePooling1D = mean over time; head 1100 sigmoid(y) + 300 (functions/optimization.py:17-19);
loss = mean |pred - y| (functions/optimization.py:47-48); Adam(1e-3) with Keras/TF defaults beta 0.9 / 0.999, epsilon 1e-7
(training_cross_validate.py:58-61) -- torch.optim.Adam's default epsilon is 1e-8; see convnet.make_optimizer.
Weights come in as a dict name -> ndarray in the PyTorch layouts (conv [out, in, k], linear [out, in])."""
import numpy as np


def conv1d_same_s2(x, w, b):
    """x [B, T, Cin] channels-last, w [Cout, Cin, 3] -> [B, ceil(T / 2), Cout]; TF SAME: pad_total = max(k - s, 0) = 1 for even T (all of
    it on the right), 2 for odd T (one each side)"""
    B, T, _ = x.shape
    if T % 2 == 0:
        xp = np.concatenate([x, np.zeros((B, 1, x.shape[2]))], axis=1)
    else:
        xp = np.concatenate([np.zeros((B, 1, x.shape[2])), x, np.zeros((B, 1, x.shape[2]))], axis=1)
    To = (T + 1) // 2
    out = np.zeros((B, To, w.shape[0]))
    for k in range(3):
        out += xp[:, k:k + 2 * To:2, :] @ w[:, :, k].T
    return out + b


def batchnorm(x, p, name, training, new_stats=None, momentum=0.99, eps=1e-3):
    axes = tuple(range(x.ndim - 1))
    if training:
        mean, var = x.mean(axis=axes), x.var(axis=axes)
        if new_stats is not None:
            new_stats[name + ".running_mean"] = momentum * p[name + ".running_mean"] + (1 - momentum) * mean
            new_stats[name + ".running_var"] = momentum * p[name + ".running_var"] + (1 - momentum) * var
    else:
        mean, var = p[name + ".running_mean"], p[name + ".running_var"]
    return (x - mean) / np.sqrt(var + eps) * p[name + ".weight"] + p[name + ".bias"]


def forward(p, x, training, new_stats=None):
    """x [B, T, 12] (already normalised) -> raw output [B]"""
    h = np.asarray(x, dtype=np.float64)
    h = np.maximum(batchnorm(conv1d_same_s2(h, p["conv1.weight"], p["conv1.bias"]), p, "bn1", training, new_stats), 0)
    h = np.maximum(batchnorm(conv1d_same_s2(h, p["conv2.weight"], p["conv2.bias"]), p, "bn2", training, new_stats), 0)
    h = conv1d_same_s2(h, p["conv3.weight"], p["conv3.bias"])      # no BN / activation on the last conv (layers.py:26)
    h = h.mean(axis=1)                                               # GlobalAveragePooling1D
    for fc, bn in (("fc1", "fbn1"), ("fc2", "fbn2"), ("fc3", "fbn3")):
        h = np.maximum(batchnorm(h @ p[fc + ".weight"].T + p[fc + ".bias"], p, bn, training, new_stats), 0)
    h = h @ p["fc4.weight"].T + p["fc4.bias"]                        # no BN / activation (layers.py:44)
    return (h @ p["out.weight"].T + p["out.bias"])[:, 0]


def predictions(raw):
    return 1100.0 / (1.0 + np.exp(-raw)) + 300.0


def mae_loss(p, x, y, training=True):
    return float(np.abs(predictions(forward(p, x, training)) - y).mean())


def adam_first_step(w, g, lr=1e-3, b1=0.9, b2=0.999, eps=1e-7):
    """the first Adam step from zero moments: m = (1 - b1) g, v = (1 - b2) g^2, bias-corrected -> w - lr g / (|g| + eps)"""
    m, v = (1 - b1) * g, (1 - b2) * g * g
    return w - lr * (m / (1 - b1)) / (np.sqrt(v / (1 - b2)) + eps)
