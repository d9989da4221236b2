"""Drop-in for the dense-`edge_index` path of the reference's Knowledge_Distillation/message_passing.py (a fork of PyG
1.6.1's MessagePassing), forward only.

  __init__ :55-80, __check_input__ :86-100, __set_size__ :115-122, __lift__ :124-136, __collect__ :138-183,
  propagate :185-261 (branch :231-261), message :263-273, aggregate :275-293, update :305-312.

Semantics kept (what a subclass can observe):
  * a `message()` parameter `name_j` / `name_i` is the `name=` argument of `propagate` gathered along the node dimension
    with `edge_index[j]` / `edge_index[i]`, where (i, j) = (1, 0) for flow 'source_to_target' and (0, 1) otherwise;
  * if `name=` is a pair `(a, b)`, `_j` ALWAYS reads `a` and `_i` ALWAYS reads `b` (:147-152 -- the element is chosen by the
    suffix, not by the flow); `a` fixes `size[0]`, `b` fixes `size[1]`, a conflicting explicit size is a ValueError (:115-122);
  * `index = edge_index[i]`, `dim_size = size[1] or size[0]` (:178-181).

The form is this repo's own: the suffix of every `message` / `aggregate` / `update` parameter is resolved ONCE in `__init__`
into (source argument, tuple element, which edge_index row) instead of being re-parsed per call, and the per-call work is
one gather per lifted argument + the HIP scatter (`tlc_scatter_f32`).  The SparseTensor / fused `message_and_aggregate`
branch (:218-228) and the TorchScript `jittable` machinery (:314-394) are not reproduced: the reference never takes them
(GATConv defines no `message_and_aggregate`; the jinja template is not shipped).
"""
import inspect

import torch

from .. import ops

_EMPTY = inspect.Parameter.empty


class MessagePassing(torch.nn.Module):
    special_args = {'edge_index', 'adj_t', 'edge_index_i', 'edge_index_j', 'size', 'size_i', 'size_j', 'ptr', 'index',
                    'dim_size'}

    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2):
        super(MessagePassing, self).__init__()
        self.aggr = aggr
        assert self.aggr in ['add', 'mean', 'max', None]
        self.flow = flow
        assert self.flow in ['source_to_target', 'target_to_source']
        self.node_dim = node_dim
        # rows of edge_index that play target (i) and source (j)
        self._row = {'_i': 1, '_j': 0} if flow == 'source_to_target' else {'_i': 0, '_j': 1}
        self._msg_params = list(inspect.signature(self.message).parameters)
        self._aggr_params = list(inspect.signature(self.aggregate).parameters)[1:]
        self._upd_params = list(inspect.signature(self.update).parameters)[1:]
        # every user argument any of the three hooks asks for, resolved once:
        #   plain name  -> (name, None, None)
        #   name_j      -> (name, 0, edge_index row of j)      element 0 of a pair, sets size[0]
        #   name_i      -> (name, 1, edge_index row of i)      element 1 of a pair, sets size[1]
        self._plan = {}
        for arg in self._msg_params + self._aggr_params + self._upd_params:
            if arg in self.special_args or arg in self._plan:
                continue
            suffix = arg[-2:]
            if suffix in self._row:
                self._plan[arg] = (arg[:-2], 0 if suffix == '_j' else 1, self._row[suffix])
            else:
                self._plan[arg] = (arg, None, None)

    def _note_size(self, size, side, t):
        """:115-122: the node count of side `side` is that of `t`; an explicit or earlier different value is an error."""
        k = t.size(self.node_dim)
        if size[side] is None:
            size[side] = k
        elif size[side] != k:
            raise ValueError('Encountered tensor with size %d in dimension %d, but expected size %d.' % (k, self.node_dim, size[side]))

    def __lift__(self, src, edge_index, dim):
        # :124-127
        return src.index_select(self.node_dim, edge_index[dim])

    def __collect__(self, args, edge_index, size, kwargs):
        """The dictionary `message` / `aggregate` / `update` draw their arguments from (:138-183)."""
        out = {}
        for arg in args:
            base, side, row = self._plan[arg] if arg in self._plan else (arg, None, None)
            val = kwargs.get(base, _EMPTY)
            if side is not None:
                if isinstance(val, (tuple, list)):
                    assert len(val) == 2
                    other = val[1 - side]
                    if isinstance(other, torch.Tensor):
                        self._note_size(size, 1 - side, other)
                    val = val[side]
                if isinstance(val, torch.Tensor):
                    self._note_size(size, side, val)
                    val = self.__lift__(val, edge_index, row)
            out[arg] = val
        tgt = edge_index[self._row['_i']]
        out.update(adj_t=None, edge_index=edge_index, edge_index_i=tgt, edge_index_j=edge_index[self._row['_j']],
                   ptr=None, index=tgt, size=size)
        out['size_i'] = size[1] or size[0]
        out['size_j'] = size[0] or size[1]
        out['dim_size'] = out['size_i']
        return out

    @staticmethod
    def _pick(names, coll):
        return {k: coll[k] for k in names if coll.get(k, _EMPTY) is not _EMPTY}

    def propagate(self, edge_index, size=None, **kwargs):
        if not isinstance(edge_index, torch.Tensor):
            raise NotImplementedError("MessagePassing (HIP): dense edge_index tensors only")
        # :89-92
        assert edge_index.dtype == torch.long and edge_index.dim() == 2 and edge_index.size(0) == 2
        if self.node_dim not in (0, -2):
            raise NotImplementedError("MessagePassing (HIP): node_dim must be 0 (or -2 for 2-D inputs)")
        size = [None, None] if size is None else [size[0], size[1]]
        coll = self.__collect__(list(self._plan), edge_index, size, kwargs)
        out = self.message(**self._pick(self._msg_params, coll))
        out = self.aggregate(out, **self._pick(self._aggr_params, coll))
        return self.update(out, **self._pick(self._upd_params, coll))

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        # :275-293  scatter(inputs, index, dim=node_dim, dim_size=dim_size, reduce=self.aggr)
        return ops.scatter(inputs, index, int(dim_size), reduce=self.aggr)

    def update(self, inputs):
        return inputs
