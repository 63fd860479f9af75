"""SharedMLP: the per-point MLP of a set-abstraction level.

Mirrors the part of /root/reference/modules/third_party/pointnet2/pytorch_utils.py
that the hot path instantiates (:11-36 SharedMLP, :39-66 BatchNorm2d wrapper,
:67-188 Conv2d), keeping the module tree -- and therefore the state-dict keys
`layer{i}.conv.weight`, `layer{i}.bn.bn.{weight,bias,running_mean,running_var,
num_batches_tracked}` -- identical so `pointnetpp.pt` checkpoints load unchanged.
"""
import torch.nn as nn


class BatchNorm2d(nn.Sequential):
    """A one-element Sequential holding `bn` (hence the `bn.bn.*` keys)."""

    def __init__(self, channels, name=""):
        super().__init__()
        norm = nn.BatchNorm2d(channels)
        nn.init.ones_(norm.weight)
        nn.init.zeros_(norm.bias)
        self.add_module(name + "bn", norm)


class Conv2d(nn.Sequential):
    """1x1 (by default) conv -> [BN] -> [activation]; bias only when there is no BN."""

    def __init__(self, in_size, out_size, *, kernel_size=(1, 1), stride=(1, 1), padding=(0, 0),
                 activation=nn.ReLU(inplace=True), bn=False, init=nn.init.kaiming_normal_,
                 bias=True, preact=False, name=""):
        super().__init__()
        use_bias = bias and not bn
        conv = nn.Conv2d(in_size, out_size, kernel_size=kernel_size, stride=stride,
                         padding=padding, bias=use_bias)
        init(conv.weight)
        if use_bias:
            nn.init.zeros_(conv.bias)
        norm = BatchNorm2d(in_size if preact else out_size) if bn else None

        def add_norm_act():
            if norm is not None:
                self.add_module(name + "bn", norm)
            if activation is not None:
                self.add_module(name + "activation", activation)

        if preact:
            add_norm_act()
        self.add_module(name + "conv", conv)
        if not preact:
            add_norm_act()


class SharedMLP(nn.Sequential):
    def __init__(self, args, *, bn=False, activation=nn.ReLU(inplace=True), preact=False,
                 first=False, name=""):
        super().__init__()
        for i in range(len(args) - 1):
            plain = first and preact and i == 0      # very first pre-activation layer is bare
            self.add_module(
                f"{name}layer{i}",
                Conv2d(args[i], args[i + 1], bn=bn and not plain,
                       activation=None if plain else activation, preact=preact))

    def forward(self, x):
        """GPU fp32 tensors: the 1x1 convolutions on this build's GEMM and the normalisation as elementwise
        operators (hipops.shared_mlp_rows) -- no MIOpen convolution / BatchNorm call anywhere on the GPU side;
        CPU tensors (the host-logic tests) take the module tree as torch runs it."""
        from .. import hipops
        if hipops.shared_mlp_rows_supported(self, x):
            return hipops.shared_mlp_rows(self, x)
        return super().forward(x)

    def conv_bn_pairs(self):
        """[(conv, bn-or-None)] per layer -- what the fused SA kernels consume."""
        out = []
        for layer in self:
            conv = bnorm = None
            for m in layer.children():
                if isinstance(m, nn.Conv2d):
                    conv = m
                elif isinstance(m, BatchNorm2d):
                    bnorm = m[0]
            out.append((conv, bnorm))
        return out
