"""Deterministic weights for the policy/value CNN, reproducible from a seed alone (numpy's legacy RandomState stream is
frozen across numpy versions), so a golden fixture of the BASELINE-shape network (2 blocks x 64 channels, 162 949
parameters) needs to store only the seed, the inputs and the reference's outputs - not 650 KB of weights.
Used by tests/golden/gen_golden.py on the REFERENCE's BasicGridNetwork and by the tests on our GridResNet."""
import numpy as np


def fill_state_dict(net, seed, gain=1.0):
    """Overwrite every tensor of net.state_dict() in key order: convolutions / linears ~ U(-b, b), b = gain / sqrt(fan_in)
    (gain 1 = the scale of PyTorch's default init; > 1 gives activations and logits of the size a trained network has),
    BatchNorm weight U(0.5, 1.5), bias N(0, 0.1), running_mean N(0, 0.2), running_var U(0.5, 1.5)."""
    import torch
    rs = np.random.RandomState(seed)
    sd = net.state_dict()
    with torch.no_grad():
        for key, t in sd.items():
            shape = tuple(t.shape)
            if key.endswith("num_batches_tracked"):
                continue
            if key.endswith("running_mean"):
                v = rs.standard_normal(shape) * 0.2
            elif key.endswith("running_var"):
                v = rs.uniform(0.5, 1.5, shape)
            elif (".bn" in key or key.startswith("bn")) and key.endswith("weight"):
                v = rs.uniform(0.5, 1.5, shape)
            elif (".bn" in key or key.startswith("bn")) and key.endswith("bias"):
                v = rs.standard_normal(shape) * 0.1
            elif key.endswith("weight"):
                fan_in = int(np.prod(shape[1:]))
                b = gain / np.sqrt(fan_in)
                v = rs.uniform(-b, b, shape)
            else:                               # conv / linear bias
                v = rs.uniform(-0.1, 0.1, shape)
            t.copy_(torch.from_numpy(np.asarray(v, np.float32)))
    return net


def othello_like_inputs(n, seed):
    """n input planes [3][8][8]: disjoint own / opponent stones at ~45 % fill, colour plane all-one or all-zero."""
    rs = np.random.RandomState(seed)
    x = np.zeros((n, 3, 8, 8), np.float32)
    for i in range(n):
        r = rs.uniform(size=(8, 8))
        fill = rs.uniform(0.1, 0.9)
        x[i, 0] = r < fill / 2
        x[i, 1] = (r >= fill / 2) & (r < fill)
        x[i, 2] = float(rs.randint(2))
    return x


def go_like_inputs(n, width, history, seed):
    """n Go input stacks [2 * history + 1][width][width] in the reference's plane order (own / opponent stones of the last
    `history` positions, then the colour plane): a random game-like sequence - stones are added one per position, older
    positions are prefixes of newer ones, a few stones disappear (captures)."""
    rs = np.random.RandomState(seed)
    P = 2 * history + 1
    x = np.zeros((n, P, width, width), np.float32)
    for i in range(n):
        cells = rs.permutation(width * width)
        nstones = int(rs.uniform(0.1, 0.7) * width * width) + history
        colour = rs.randint(2, size=width * width)
        for h in range(history):                   # position h plies ago holds the first nstones - h stones
            own = np.zeros(width * width, np.float32)
            opp = np.zeros(width * width, np.float32)
            live = cells[:max(0, nstones - h)]
            live = live[rs.uniform(size=len(live)) > 0.03]         # a few captured
            side = colour[live] ^ (h & 1)
            own[live[side == 0]] = 1.0
            opp[live[side == 1]] = 1.0
            x[i, 2 * h] = own.reshape(width, width)
            x[i, 2 * h + 1] = opp.reshape(width, width)
        x[i, P - 1] = float(rs.randint(2))
    return x
