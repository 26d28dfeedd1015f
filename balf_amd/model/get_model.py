"""Drop-in for ``balf.model.get_model`` (/root/reference/balf/model/get_model.py): same three
functions, same arguments, same return values and the same exceptions."""
from __future__ import annotations

import os

import torch

from . import mlp_ma_decoder


def _load_into(model, checkpoint, say):
    disk = checkpoint['model_state']
    own = model.state_dict()
    update = {k: v for k, v in disk.items() if k in own and own[k].shape == v.shape}
    for k, v in update.items():
        say('Update weight %s: %s' % (k, str(v.shape)), True)
    own.update(update)
    model.load_state_dict(own)
    for k in own:
        if k not in update:
            say('Not updated weight %s: %s' % (k, str(own[k].shape)), False)
    return update, own


def _load_optimizer(optimizer, checkpoint, filename, loc):
    if optimizer is None:
        return
    if checkpoint.get('optimizer_state') is not None:
        optimizer.load_state_dict(checkpoint['optimizer_state'])
        return
    assert filename[-4] == '.', filename
    side = '%s_optim.%s' % (filename[:-4], filename[-3:])
    if os.path.exists(side):
        optimizer.load_state_dict(torch.load(side, map_location=loc)['optimizer_state'])     # get_model.py:44,80


def load_test_pretrained_model(model, filename, optimizer=None, device='cuda'):
    """get_model.py:50-86.  Returns (epoch, repeatability); FileNotFoundError if the file is
    missing; AssertionError unless every state-dict entry was matched by name and shape."""
    if not os.path.isfile(filename):
        raise FileNotFoundError
    loc = torch.device('cpu') if device == 'cpu' else None
    checkpoint = torch.load(filename, map_location=loc, weights_only=False)
    update, own = _load_into(model, checkpoint, lambda msg, updated: None if updated else print(msg))
    _load_optimizer(optimizer, checkpoint, filename, loc)
    assert len(update) == len(own)
    return checkpoint.get('epoch', -1), checkpoint.get('repeatability', 0.0)


def load_pretrained_model(model, filename, logger, optimizer=None, device='cuda'):
    """get_model.py:6-48: as above, reporting through ``logger.info``."""
    if not os.path.isfile(filename):
        raise FileNotFoundError
    logger.info('==> Loading parameters from checkpoint %s to %s' % (filename, 'CPU' if device == 'cpu' else 'GPU'))
    loc = torch.device('cpu') if device == 'cpu' else None
    checkpoint = torch.load(filename, map_location=loc)      # torch's default unpickling policy, as get_model.py:14-15
    update, own = _load_into(model, checkpoint, lambda msg, updated: logger.info(msg))
    _load_optimizer(optimizer, checkpoint, filename, loc)
    assert len(update) == len(own)
    logger.info('==> Done (loaded %d/%d)' % (len(update), len(own)))
    return checkpoint.get('epoch', -1), checkpoint.get('repeatability', 0.0)


def load_model(model_cfg):
    """get_model.py:88-90; ``model_cfg`` is ``cfg['model']`` of the reference's YAML."""
    return mlp_ma_decoder.MLP_MA_DECODER(model_cfg['network_architecture'])
