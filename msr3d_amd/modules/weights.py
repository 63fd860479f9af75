"""Parameter initialisation of the trainable part: the BERT scheme the reference applies to
`obj_linear_projection`, the spatial encoder and the location encoders
(/root/reference/modules/weights.py:3-19 via model/ose3d_situation.py:193-194,250-251):
N(0, std) matrices, zero biases, unit LayerNorm gains, zero row for an embedding's padding index."""
import torch
import torch.nn as nn


@torch.no_grad()
def _init_weights_bert(module, std=0.02):
    """For `nn.Module.apply`: touches Linear, Embedding and LayerNorm, leaves the rest alone."""
    if isinstance(module, nn.LayerNorm):
        module.weight.fill_(1.0)
        module.bias.zero_()
        return
    if not isinstance(module, (nn.Linear, nn.Embedding)):
        return
    module.weight.normal_(0.0, std)
    bias = getattr(module, "bias", None)
    if bias is not None:
        bias.zero_()
    pad = getattr(module, "padding_idx", None)
    if pad is not None:
        module.weight[pad].zero_()
