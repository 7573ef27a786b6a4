"""Checkpoint migration -- drop-in for src/waveglow/convert_model.py: models saved before the
res/skip 1x1 convolutions were merged carry ``res_layers`` and ``skip_layers`` per WN; fold them into
the ``res_skip_layers`` the kernels (and the current module layout) expect.  Load-time only."""
import io
import sys

import torch


def _clone(model):
    """Deep copy through the checkpoint pickle path: copy.deepcopy refuses weight-normed modules on
    current PyTorch (their ``weight`` attribute is a non-leaf tensor), torch.save/load does not."""
    buf = io.BytesIO()
    torch.save(model, buf)
    buf.seek(0)
    return torch.load(buf, weights_only=False)


def _check_model_old_version(model):
    return hasattr(model.WN[0], 'res_layers')


def update_model(old_model):
    """convert_model.py:43-70"""
    if not _check_model_old_version(old_model):
        return old_model
    new_model = _clone(old_model)
    for wavenet in new_model.WN:
        merged = torch.nn.ModuleList()
        for i in range(wavenet.n_layers):
            skip = torch.nn.utils.remove_weight_norm(wavenet.skip_layers[i])
            parts_w, parts_b = [skip.weight], [skip.bias]
            if i < wavenet.n_layers - 1:
                res = torch.nn.utils.remove_weight_norm(wavenet.res_layers[i])
                parts_w, parts_b = [res.weight, skip.weight], [res.bias, skip.bias]
            conv = torch.nn.Conv1d(wavenet.n_channels, sum(p.shape[0] for p in parts_w), 1)
            conv.weight = torch.nn.Parameter(torch.cat(parts_w))
            conv.bias = torch.nn.Parameter(torch.cat(parts_b))
            merged.append(torch.nn.utils.weight_norm(conv, name='weight'))
        wavenet.res_skip_layers = merged
        del wavenet.res_layers
        del wavenet.skip_layers
    return new_model


if __name__ == '__main__':
    model = torch.load(sys.argv[1], weights_only=False)
    model['model'] = update_model(model['model'])
    torch.save(model, sys.argv[2])
