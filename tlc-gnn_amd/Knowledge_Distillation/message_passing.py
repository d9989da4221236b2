"""Drop-in for the dense-`edge_index` path of the reference's Knowledge_Distillation/message_passing.py (a fork of PyG
1.6.1's MessagePassing), forward only.

  __init__ :55-80, __lift__ :124-136, __collect__ :138-183, propagate :185-261 (branch :231-261), message :263-273,
  aggregate :275-293, update :305-312.

`propagate` gathers the `_j` / `_i` arguments of `message()` (index_select on the device), calls the subclass's `message`,
aggregates at the target with the HIP scatter (`tlc_scatter_f32`) and calls `update`.  The SparseTensor / fused
`message_and_aggregate` branch (:218-228) and the TorchScript `jittable` machinery (:314-394) are not reproduced: the
reference never takes them (GATConv defines no `message_and_aggregate`; the jinja template is not shipped).
"""
import inspect

import torch

from .. import ops


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
        self._msg_params = [p for p in inspect.signature(self.message).parameters]
        self._upd_params = [p for p in list(inspect.signature(self.update).parameters)[1:]]

    def __lift__(self, src, edge_index, dim):
        # :124-127  index_select along the node dimension
        return src.index_select(self.node_dim, edge_index[dim])

    def __collect__(self, args, edge_index, size, kwargs):
        # :138-183 for a dense edge_index: j = source row, i = target row for 'source_to_target'
        i, j = (1, 0) if self.flow == 'source_to_target' else (0, 1)
        out = {}
        for arg in args:
            if arg[-2:] not in ['_i', '_j']:
                out[arg] = kwargs.get(arg, inspect.Parameter.empty)
            else:
                dim = 0 if arg[-2:] == '_j' else 1
                data = kwargs.get(arg[:-2], inspect.Parameter.empty)
                if isinstance(data, (tuple, list)):
                    assert len(data) == 2
                    data = data[1 - dim]
                if isinstance(data, torch.Tensor):
                    if size[1 - dim] is None:
                        size[1 - dim] = data.size(self.node_dim)
                    data = self.__lift__(data, edge_index, j if arg[-2:] == '_j' else i)
                out[arg] = data
        out['adj_t'] = None
        out['edge_index'] = edge_index
        out['edge_index_i'] = edge_index[i]
        out['edge_index_j'] = edge_index[j]
        out['ptr'] = None
        out['index'] = out['edge_index_i']
        out['size'] = size
        out['size_i'] = size[1] if size[1] is not None else size[0]
        out['size_j'] = size[0] if size[0] is not None else size[1]
        out['dim_size'] = out['size_i']
        return out

    def propagate(self, edge_index, size=None, **kwargs):
        if not isinstance(edge_index, torch.Tensor):
            raise NotImplementedError("MessagePassing (HIP): dense edge_index tensors only")
        if self.node_dim not in (0, -2):
            raise NotImplementedError("MessagePassing (HIP): node_dim must be 0 (or -2 for 2-D inputs)")
        size = [None, None] if size is None else list(size)
        coll = self.__collect__(self._msg_params, edge_index, size, kwargs)
        msg_kwargs = {k: coll[k] for k in self._msg_params if coll.get(k, inspect.Parameter.empty) is not inspect.Parameter.empty}
        out = self.message(**msg_kwargs)
        aggr_params = [p for p in list(inspect.signature(self.aggregate).parameters)[1:]]
        aggr_kwargs = {k: coll[k] for k in aggr_params if k in coll}
        out = self.aggregate(out, **aggr_kwargs)
        upd_kwargs = {k: coll.get(k, kwargs.get(k)) for k in self._upd_params if (k in coll or k in kwargs)}
        return self.update(out, **upd_kwargs)

    def message(self, x_j):
        return x_j

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        # :275-293  scatter(inputs, index, dim=node_dim, dim_size=dim_size, reduce=self.aggr)
        return ops.scatter(inputs, index, int(dim_size), reduce=self.aggr)

    def update(self, inputs):
        return inputs
