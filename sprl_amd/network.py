"""Policy/value CNN consumed by the self-play engine — the *contract* side of the reference's
`src/networks/grid_networks.py:30-79` (architecture) and `src/interface/tracer.py:10-19` (export).

The engine only ever sees a TorchScript file whose forward maps float32[B, 2H+1, R, C] to
(logits float32[B, A], value float32[B, 1]) (`cpp/src/networks/GridNetwork.hpp:99-102`).  This module is
our own definition of that network so that bench.py / tests can produce random-init traced models of the
BASELINE shape (Othello: 3 -> 64 channels, 2 residual blocks).  Parameter names follow the reference's
state_dict keys so checkpoints trained by the reference controller load unchanged
(pinned by tests/golden/g9_network.npz).
"""
from typing import Tuple

import torch
from torch import nn
import torch.nn.functional as F


class _Residual(nn.Module):
    """3x3 conv + BN + ReLU, 3x3 conv + BN, skip add, ReLU (grid_networks.py:8-27)."""

    def __init__(self, channels: int):
        super().__init__()
        self.conv1 = nn.Conv2d(channels, channels, 3, 1, 1)
        self.bn1 = nn.BatchNorm2d(channels)
        self.conv2 = nn.Conv2d(channels, channels, 3, 1, 1)
        self.bn2 = nn.BatchNorm2d(channels)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        y = F.relu(self.bn1(self.conv1(x)))
        y = self.bn2(self.conv2(y))
        return F.relu(y + x)


class GridResNet(nn.Module):
    """Stem -> `num_blocks` residual blocks -> (1x1 policy head -> FC(A), 1x1 value head -> FC -> FC -> tanh)."""

    def __init__(self, num_rows: int, num_cols: int, action_size: int, history_size: int = 1,
                 num_blocks: int = 2, num_channels: int = 64, num_policy_channels: int = 2,
                 num_value_channels: int = 1):
        super().__init__()
        cells = num_rows * num_cols
        self.conv = nn.Conv2d(2 * history_size + 1, num_channels, 3, 1, 1)
        self.bn = nn.BatchNorm2d(num_channels)
        self.residual_blocks = nn.ModuleList(_Residual(num_channels) for _ in range(num_blocks))
        self.policy_conv = nn.Conv2d(num_channels, num_policy_channels, 1)
        self.policy_fc = nn.Linear(num_policy_channels * cells, action_size)
        self.value_conv = nn.Conv2d(num_channels, num_value_channels, 1)
        self.value_fc1 = nn.Linear(num_value_channels * cells, num_channels)
        self.value_fc2 = nn.Linear(num_channels, 1)

    def forward(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        x = F.relu(self.bn(self.conv(x)))
        for block in self.residual_blocks:
            x = block(x)
        p = F.relu(self.policy_conv(x)).flatten(1)
        v = F.relu(self.value_conv(x)).flatten(1)
        logits = self.policy_fc(p)
        value = torch.tanh(self.value_fc2(F.relu(self.value_fc1(v))))
        return logits, value


GAME_SHAPES = {
    # game: (rows, cols, actions, history)  — OthelloNode.hpp:8-11, ConnectFourNode.hpp:8-13
    "othello": (8, 8, 65, 1),
    "connect_four": (6, 7, 7, 1),
    "go7": (7, 7, 50, 8),                 # GoNode.hpp:16-19
    "go9": (9, 9, 82, 8),
    "go19": (19, 19, 362, 8),
}


def make_network(game: str, num_blocks: int = 2, num_channels: int = 64, seed: int = 0) -> GridResNet:
    """Random-init network of the BASELINE shape for `game` (eval mode, float32, CPU)."""
    rows, cols, actions, hist = GAME_SHAPES[game]
    gen_state = torch.random.get_rng_state()
    torch.manual_seed(seed)
    net = GridResNet(rows, cols, actions, hist, num_blocks, num_channels)
    # random-init BatchNorm statistics too, so eval-mode BN is not an identity
    with torch.no_grad():
        for m in net.modules():
            if isinstance(m, nn.BatchNorm2d):
                m.running_mean.normal_(0.0, 0.1)
                m.running_var.uniform_(0.5, 1.5)
                m.weight.uniform_(0.5, 1.5)
                m.bias.normal_(0.0, 0.1)
    torch.random.set_rng_state(gen_state)
    return net.eval()


def _trace(net: nn.Module, game: str):
    rows, cols, _, hist = GAME_SHAPES[game]
    example = torch.randn(1, 2 * hist + 1, rows, cols)
    with torch.no_grad():
        return torch.jit.trace(net.cpu().eval(), example)


def trace_to_file(net: nn.Module, path: str, game: str) -> str:
    """`torch.jit.trace` + save, like the reference controller does after each training iteration
    (scripts/othello_controller.py:237-239, tracer.py:10-19): CPU weights, example randn(1, 2H+1, R, C)."""
    _trace(net, game).save(path)
    return path


def trace_to_bytes(net: nn.Module, game: str) -> bytes:
    """The same TorchScript archive in memory (for sprl_engine_set_model_buffer): no file is written or read."""
    import io
    buf = io.BytesIO()
    torch.jit.save(_trace(net, game), buf)
    return buf.getvalue()
