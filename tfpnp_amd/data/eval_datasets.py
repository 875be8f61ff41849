"""On-disk evaluation items -- the `.mat` format of the reference's *EvalDataset classes
(tasks/csmri/dataset.py:86-109, tasks/spi/dataset.py:75-98): one MATLAB v5 file per sample with the keys the
generator scripts wrote (CS-MRI: y0, x0, ATy0, gt, mask, sigma_n, name; SPI: x0, gt, K, name).  `save_eval_item`
writes the same layout, so items synthesised on the GPU (data/synthesis.py) can be frozen as fixed test sets."""
import os

import numpy as np
import torch
from scipy.io import loadmat, savemat


def _strip(mat):
    for k in ('__globals__', '__header__', '__version__'):
        mat.pop(k, None)
    mat['name'] = np.asarray(mat['name']).item() if 'name' in mat else ''
    return mat


class CSMRIEvalDataset:
    """tasks/csmri/dataset.py:86-109.  Items: y0/x0/ATy0/sigma_n [1,H,W,2], gt [1,H,W], mask bool [1,H,W]."""

    def __init__(self, datadir, fns=None):
        self.datadir = datadir
        self.fns = sorted(fns or [f for f in os.listdir(datadir) if f.endswith('.mat')])

    def __getitem__(self, index):
        mat = _strip(loadmat(os.path.join(self.datadir, self.fns[index])))
        mat['output'] = mat['ATy0'][..., 0]
        mat['input'] = mat['x0']
        mat['mask'] = np.expand_dims(mat['mask'], axis=0).astype('bool')
        return mat

    def __len__(self):
        return len(self.fns)


class SPIEvalDataset:
    """tasks/spi/dataset.py:75-98.  Items: x0, gt [1,H,W]; K scalar -> constant map K/10."""

    def __init__(self, datadir, fns=None):
        self.datadir = datadir
        self.fns = sorted(fns or [f for f in os.listdir(datadir) if f.endswith('.mat')])

    def __getitem__(self, index):
        mat = _strip(loadmat(os.path.join(self.datadir, self.fns[index])))
        mat['output'] = mat['x0']
        mat['input'] = mat['x0']
        mat['K'] = (np.ones_like(mat['gt']) * np.asarray(mat['K']).reshape(1, 1, 1) / 10.).astype(np.float32)
        return mat

    def __len__(self):
        return len(self.fns)


def save_eval_item(path, item, task='csmri'):
    """Write one sample (tensors / arrays WITHOUT the batch dim) in the reference's .mat layout."""
    def a(v):
        return v.detach().cpu().numpy() if isinstance(v, torch.Tensor) else np.asarray(v)
    if task == 'csmri':
        out = {k: a(item[k]) for k in ('y0', 'x0', 'ATy0', 'gt', 'sigma_n')}
        out['mask'] = a(item['mask']).astype(np.uint8).reshape(a(item['mask']).shape[-2:])
    elif task == 'spi':
        out = {'x0': a(item['x0']), 'gt': a(item['gt']), 'K': np.asarray(item['K'], np.float32).reshape(-1)[:1] * 1.0}
    else:
        raise NotImplementedError(task)
    out['name'] = item.get('name', os.path.splitext(os.path.basename(path))[0])
    savemat(path, out)


def collate(items, device=None):
    """batch-1 (or batch-N) dict of device tensors from dataset items; 'name' stays a list."""
    out = {}
    for k in items[0]:
        if k == 'name':
            out[k] = [it[k] for it in items]
            continue
        v = torch.from_numpy(np.stack([np.asarray(it[k]) for it in items]))
        out[k] = v.to(device) if device is not None else v
    return out
