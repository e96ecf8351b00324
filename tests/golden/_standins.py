"""Stand-ins for the third-party packages the reference imports but this image lacks.

Used ONLY by tests/golden/make_golden.py, in the build container, to import the
reference's own ``infgen.modules.*`` from /root/reference and run it on CPU.  Contains no
reference code: it re-states the *published* semantics of the pinned third-party ops
(environment.yml: torch-cluster 1.6.3, torch-geometric 2.5.3):

* ``torch_cluster.radius(x, y, r, batch_x, batch_y, max_num_neighbors)`` -> ``[y_idx; x_idx]``,
  for each ``y`` the first K ``x`` rows (ascending index, the CUDA kernel's order) of the
  same batch id with squared distance strictly below ``r*r``.
* ``radius_graph(x, r, batch, loop, K)`` = ``radius(x, x, r, batch, batch, K (+1 if not loop))``
  flipped to ``[src; dst]`` with self loops dropped.
* PyG ``MessagePassing.propagate`` (aggr='add', node_dim=0), ``utils.softmax`` (max-shift,
  ``/(sum + 1e-16)``), ``dense_to_sparse`` (3-D), ``subgraph`` (bool mask).
"""
from __future__ import annotations

import copy
import inspect
import os
import sys
import types
from unittest import mock

import torch
import torch.nn as nn


def _radius(x, y, r, batch_x=None, batch_y=None, max_num_neighbors=32, num_workers=1, batch_size=None):
    if batch_x is None:
        batch_x = torch.zeros(x.shape[0], dtype=torch.long)
    if batch_y is None:
        batch_y = torch.zeros(y.shape[0], dtype=torch.long)
    rows, cols = [], []
    r2 = float(r) * float(r)
    for b in torch.unique(batch_y).tolist():
        yi = torch.nonzero(batch_y == b)[:, 0]
        xi = torch.nonzero(batch_x == b)[:, 0]
        if yi.numel() == 0 or xi.numel() == 0:
            continue
        d = ((y[yi][:, None, :] - x[xi][None, :, :]) ** 2).sum(-1)
        within = d < r2
        rank = torch.cumsum(within.long(), dim=1)
        keep = within & (rank <= max_num_neighbors)
        nz = torch.nonzero(keep)
        rows.append(yi[nz[:, 0]])
        cols.append(xi[nz[:, 1]])
    if not rows:
        return torch.zeros(2, 0, dtype=torch.long)
    row = torch.cat(rows)
    col = torch.cat(cols)
    order = torch.argsort(row * (x.shape[0] + 1) + col)
    return torch.stack([row[order], col[order]])


def _radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow='source_to_target', num_workers=1,
                  batch_size=None):
    ei = _radius(x, x, r, batch, batch, max_num_neighbors if loop else max_num_neighbors + 1)
    if flow == 'source_to_target':
        row, col = ei[1], ei[0]
    else:
        row, col = ei[0], ei[1]
    if not loop:
        m = row != col
        row, col = row[m], col[m]
    return torch.stack([row, col])


def _softmax(src, index, ptr=None, num_nodes=None, dim=0):
    n = int(index.max()) + 1 if num_nodes is None and index.numel() > 0 else (num_nodes or 0)
    shape = (n,) + tuple(src.shape[1:])
    idx = index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)
    mx = torch.full(shape, float('-inf'), dtype=src.dtype).scatter_reduce(0, idx, src, reduce='amax', include_self=True)
    out = (src - mx.gather(0, idx)).exp()
    sm = torch.zeros(shape, dtype=src.dtype).scatter_add(0, idx, out)
    return out / (sm.gather(0, idx) + 1e-16)


def _dense_to_sparse(adj, mask=None):
    if adj.dim() == 2:
        idx = adj.nonzero().t()
        return idx, adj[idx[0], idx[1]]
    nz = adj.nonzero()
    b, i, j = nz[:, 0], nz[:, 1], nz[:, 2]
    n = adj.shape[1]
    return torch.stack([b * n + i, b * adj.shape[2] + j]), adj[b, i, j]


def _subgraph(subset, edge_index, edge_attr=None, relabel_nodes=False, num_nodes=None, return_edge_mask=False):
    assert subset.dtype == torch.bool
    m = subset[edge_index[0]] & subset[edge_index[1]]
    return edge_index[:, m], (edge_attr[m] if edge_attr is not None else None)


class MessagePassing(nn.Module):
    def __init__(self, aggr='add', node_dim=0, **kwargs):
        super().__init__()
        assert aggr == 'add' and node_dim == 0

    def propagate(self, edge_index, size=None, **kwargs):
        msg_params = list(inspect.signature(self.message).parameters)
        upd_params = list(inspect.signature(self.update).parameters)
        src, dst = edge_index[0], edge_index[1]
        margs = {}
        for p in msg_params:
            if p.endswith('_i'):
                margs[p] = kwargs[p[:-2]][dst]
            elif p.endswith('_j'):
                margs[p] = kwargs[p[:-2]][src]
            elif p == 'index':
                margs[p] = dst
            elif p == 'ptr':
                margs[p] = None
            else:
                margs[p] = kwargs.get(p)
        n_dst = kwargs['q'].size(0)
        if 'index' in margs and margs['index'].numel() == 0:
            # empty edge set: softmax over nothing
            pass
        self._n_dst = n_dst
        msg = self.message(**margs)
        out = torch.zeros((n_dst,) + tuple(msg.shape[1:]), dtype=msg.dtype)
        out.index_add_(0, dst, msg)
        uargs = {p: kwargs[p] for p in upd_params[1:] if p in kwargs}
        return self.update(out, **uargs)


class HeteroData(dict):
    """dict with ``num_graphs`` and a deep-copy ``clone``; keys may be tuples."""
    num_graphs = 1

    def clone(self):
        return copy.deepcopy(self)


class Batch(HeteroData):
    pass


def _softmax_safe(src, index, ptr=None, num_nodes=None, dim=0):
    if index.numel() == 0:
        return src
    return _softmax(src, index, ptr, num_nodes, dim)


REFERENCE = '/root/reference'


def bind_reference():
    """Make the name ``infgen`` mean the REFERENCE, whatever else is importable: the reference's ``infgen/`` has no
    ``__init__.py`` (a namespace package), so any regular package called ``infgen`` on ``sys.path`` - e.g. this repository's
    ``compat/infgen`` alias over ``infgen_amd`` - would win over it regardless of the path order.  A module object whose
    ``__path__`` is the reference's directory alone is registered before anything imports ``infgen``; a previously imported
    ``infgen`` from anywhere else is an error (the fixtures would be produced by the code under test)."""
    import importlib.machinery
    import types
    ref_pkg = os.path.join(REFERENCE, 'infgen')
    if not os.path.isdir(ref_pkg):
        raise RuntimeError(f'{ref_pkg} not found: the golden generators run in the build container only')
    have = sys.modules.get('infgen')
    if have is not None:
        if list(getattr(have, '__path__', [])) != [ref_pkg]:
            raise RuntimeError(f'`infgen` is already imported from {getattr(have, "__path__", None)}, not from the reference')
        return
    if REFERENCE not in sys.path:
        sys.path.insert(0, REFERENCE)
    m = types.ModuleType('infgen')
    m.__path__ = [ref_pkg]
    m.__spec__ = importlib.machinery.ModuleSpec('infgen', None, is_package=True)
    m.__spec__.submodule_search_locations = [ref_pkg]
    sys.modules['infgen'] = m


def assert_reference(obj):
    """``obj`` (class / function / module) was defined by a file under /root/reference"""
    import inspect
    f = inspect.getsourcefile(obj) or ''
    if not os.path.realpath(f).startswith(REFERENCE + os.sep):
        raise RuntimeError(f'{obj!r} comes from {f}, not from the reference')
    return obj


def install(bind: bool = True):
    """Register the stand-in modules in sys.modules (idempotent) and bind ``infgen`` to the reference (bind=False: the
    third-party stand-ins only - tests/test_compat_entry_cpu.py imports the reference's entry scripts against compat/infgen)."""
    if bind:
        bind_reference()
    if 'torch_cluster' in sys.modules and getattr(sys.modules['torch_cluster'], '_is_standin', False):
        return

    def fake(name):
        m = mock.MagicMock()
        m.__path__ = []
        m.__name__ = name
        m.__all__ = []
        m.__spec__ = None
        return m

    names = [
        'torch_geometric', 'torch_geometric.data', 'torch_geometric.utils', 'torch_geometric.nn',
        'torch_geometric.nn.conv', 'torch_geometric.loader', 'torch_geometric.transforms',
        'torch_cluster', 'torch_scatter', 'torchmetrics',
        'pytorch_lightning', 'pytorch_lightning.callbacks', 'pytorch_lightning.strategies',
        'pytorch_lightning.loggers',
        'easydict', 'lightning_utilities', 'lightning_utilities.core', 'lightning_utilities.core.rank_zero',
        'tensorflow', 'seaborn',
        'waymo_open_dataset', 'waymo_open_dataset.protos', 'waymo_open_dataset.protos.scenario_pb2',
        'waymo_open_dataset.utils', 'waymo_open_dataset.utils.sim_agents',
        'waymo_open_dataset.utils.sim_agents.submission_specs',
        'scipy.ndimage.filters',
    ]
    for n in names:
        if n not in sys.modules:
            sys.modules[n] = fake(n)

    tc = sys.modules['torch_cluster']
    tc._is_standin = True
    tc.radius = _radius
    tc.radius_graph = _radius_graph

    conv = sys.modules['torch_geometric.nn.conv']
    conv.MessagePassing = MessagePassing
    sys.modules['torch_geometric.nn'].conv = conv
    sys.modules['torch_geometric.nn'].MessagePassing = MessagePassing

    utils = sys.modules['torch_geometric.utils']
    utils.softmax = _softmax_safe
    utils.dense_to_sparse = _dense_to_sparse
    utils.subgraph = _subgraph

    data = sys.modules['torch_geometric.data']
    data.HeteroData = HeteroData
    data.Batch = Batch
    data.Dataset = type('Dataset', (), {})
    sys.modules['torch_geometric.transforms'].BaseTransform = object

    tm = sys.modules['torchmetrics']

    class Metric(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

        def add_state(self, *a, **k):
            pass
    tm.Metric = Metric

    pl = sys.modules['pytorch_lightning']

    class LightningModule(nn.Module):
        def save_hyperparameters(self, *a, **k):
            pass
    pl.LightningModule = LightningModule
    pl.LightningDataModule = object

    class EasyDict(dict):
        def __init__(self, d=None, **kw):
            super().__init__()
            d = dict(d or {}, **kw)
            for k, v in d.items():
                self[k] = EasyDict(v) if isinstance(v, dict) else v

        __getattr__ = dict.__getitem__
        __setattr__ = dict.__setitem__
    sys.modules['easydict'].EasyDict = EasyDict

    rz = sys.modules['lightning_utilities.core.rank_zero']
    rz.rank_prefixed_message = lambda msg, rank=None: msg
    rz.rank_zero_only = lambda f: f
