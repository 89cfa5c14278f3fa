"""ConvNet stiffness regressor of the reference (net/NeuralNets.py:6-27, net/layers.py:13-52) in PyTorch-ROCm,
for BASELINE.json configs[4]: it consumes the simulator's on-device ``[n, 200, 12]`` observation block directly.

Layer by layer: Conv1D(128,3,s2,SAME)+BN+ReLU, Conv1D(256,3,s2,SAME)+BN+ReLU, Conv1D(512,3,s2,SAME),
GlobalAveragePooling1D, Dense 512/256/128 (+BN+ReLU each), Dense 64, Dense 1.  Keras defaults are mirrored:
Glorot-uniform kernels, zero bias, BatchNorm momentum 0.99 and eps 1e-3 with Keras' update rule (KerasBatchNorm); TF's SAME padding for
kernel 3 / stride 2 on an even length pads (0, 1).  Head: ``1100 * sigmoid(y) + 300`` (functions/optimization.py:17-19).
Input normalisation ``(x - mean) / std`` over axes (0, 1) (functions/utils.py:40-41); noise augmentation sigma 0.7 on the
accelerometer channels 0:6 and 0.06 on the gyro channels 6:12 (functions/optimization.py:6-14).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class KerasBatchNorm(nn.Module):
    """tf.keras.layers.BatchNormalization with its defaults (net/layers.py:27,46: momentum 0.99, epsilon 1e-3, gamma 1, beta 0) over
    the channel axis 1 of ``[B, C]`` or ``[B, C, T]``.  Training: normalise with the batch mean and the POPULATION variance and
    move the statistics by ``moving = 0.99 moving + 0.01 batch`` -- the moving variance too takes the population variance
    (torch.nn.BatchNorm1d feeds it the unbiased one: a 1 % difference at the reference's batch size 100 on the dense layers).
    Inference: the moving statistics."""

    def __init__(self, channels, momentum=0.99, eps=1e-3):
        super().__init__()
        self.momentum, self.eps = momentum, eps
        self.weight = nn.Parameter(torch.ones(channels))
        self.bias = nn.Parameter(torch.zeros(channels))
        self.register_buffer("running_mean", torch.zeros(channels))
        self.register_buffer("running_var", torch.ones(channels))

    def forward(self, x):
        dims = (0,) if x.dim() == 2 else (0, 2)
        shape = (1, -1) if x.dim() == 2 else (1, -1, 1)
        if self.training:
            mean = x.mean(dim=dims)
            var = x.var(dim=dims, unbiased=False)
            with torch.no_grad():
                self.running_mean.mul_(self.momentum).add_(mean.detach(), alpha=1 - self.momentum)
                self.running_var.mul_(self.momentum).add_(var.detach(), alpha=1 - self.momentum)
        else:
            mean, var = self.running_mean, self.running_var
        return (x - mean.view(shape)) * torch.rsqrt(var.view(shape) + self.eps) * self.weight.view(shape) + self.bias.view(shape)


class ConvNet(nn.Module):
    def __init__(self, in_channels=12):
        super().__init__()
        self.conv1, self.bn1 = nn.Conv1d(in_channels, 128, 3, stride=2), KerasBatchNorm(128)
        self.conv2, self.bn2 = nn.Conv1d(128, 256, 3, stride=2), KerasBatchNorm(256)
        self.conv3 = nn.Conv1d(256, 512, 3, stride=2)
        self.fc1, self.fbn1 = nn.Linear(512, 512), KerasBatchNorm(512)
        self.fc2, self.fbn2 = nn.Linear(512, 256), KerasBatchNorm(256)
        self.fc3, self.fbn3 = nn.Linear(256, 128), KerasBatchNorm(128)
        self.fc4 = nn.Linear(128, 64)
        self.out = nn.Linear(64, 1)
        for m in self.modules():
            if isinstance(m, (nn.Conv1d, nn.Linear)):
                nn.init.xavier_uniform_(m.weight)   # Glorot uniform, as Keras
                nn.init.zeros_(m.bias)

    @staticmethod
    def _same(x):  # TF "SAME" for kernel 3, stride 2: total padding 1 when the length is even, all of it on the right
        return F.pad(x, (0, 1)) if x.shape[-1] % 2 == 0 else F.pad(x, (1, 1))

    def forward(self, x):
        """x: [B, T, 12] (channels last, any float dtype) -> raw output [B, 1]"""
        # tf.cast(inputs, tf.float32) (NeuralNets.py:22): the network computes in the dtype of its parameters -- float32 as built;
        # a .double() copy is the fp64 evaluation the numeric tests hold the float32 one against
        x = x.to(self.conv1.weight.dtype).transpose(1, 2)
        x = F.relu(self.bn1(self.conv1(self._same(x))))
        x = F.relu(self.bn2(self.conv2(self._same(x))))
        x = self.conv3(self._same(x))                       # no BN / activation on the last conv (layers.py:26)
        x = x.mean(dim=-1)                                  # GlobalAveragePooling1D
        x = F.relu(self.fbn1(self.fc1(x)))
        x = F.relu(self.fbn2(self.fc2(x)))
        x = F.relu(self.fbn3(self.fc3(x)))
        x = self.fc4(x)                                     # no BN / activation (layers.py:44)
        return self.out(x)


def normalize_predictions(preds):
    """functions/optimization.py:17-19"""
    return (1100.0 * torch.sigmoid(preds) + 300.0).squeeze(-1)


def channel_stats(x):
    """per-channel mean / std over axes (0, 1), keepdims (functions/utils.py:40-41; np.std = population std)"""
    return x.mean(dim=(0, 1), keepdim=True), x.std(dim=(0, 1), keepdim=True, unbiased=False)


def noised_modality(x, generator=None):
    """functions/optimization.py:6-14"""
    noise = torch.randn(x.shape, dtype=x.dtype, device=x.device, generator=generator)
    scale = torch.cat([torch.full((6,), 0.7), torch.full((x.shape[-1] - 6,), 0.06)]).to(x)
    return x + noise * scale


def make_optimizer(model, lr=1e-3):
    """tf.keras.optimizers.Adam(learning_rate) as the reference builds it (training_cross_validate.py:58-61; the ExponentialDecay there is
    always evaluated at step 0, so the rate is a constant 1e-3): beta 0.9 / 0.999 and Keras' epsilon 1e-7 (torch's default is 1e-8)"""
    return torch.optim.Adam(model.parameters(), lr=lr, betas=(0.9, 0.999), eps=1e-7)


def train_step(model, optimizer, x, y, mean, std, add_noise=False):
    """one eager step of functions/optimization.py:31-51: MAE on the rescaled sigmoid head (the L2 term there is computed
    and then overwritten, so the gradients carry no weight decay)"""
    model.train()
    if add_noise:
        x = noised_modality(x)
    pred = normalize_predictions(model((x - mean) / std))
    loss = (pred - y.to(pred.dtype)).abs().mean()
    optimizer.zero_grad(set_to_none=True)
    loss.backward()
    optimizer.step()
    return loss.detach(), pred.detach()
