"""Autograd glue for PDGNN training: `loss.backward()` of the reference's training loop
(Knowledge_Distillation/train_Teacher_Model.py:55-62) through the HIP kernels.

torch.autograd only carries the graph: every forward and every backward below is one C-ABI call (`tlc_gat_layer_fwd/_bwd`,
`tlc_edge_head_fwd/_bwd`, `tlc_w2_partial_matching`, `tlc_pi_raster` / `tlc_pi_raster_wgrad`); nothing is recomputed with torch ops and there is no CPU path.
"""
import torch

from . import ops, engine


class GatLayer(torch.autograd.Function):
    """One PDGNN layer (gat_conv.py:113-216) with the PReLU that follows it in Base_Model.forward (Teacher_model.py:218-219)."""

    @staticmethod
    def forward(ctx, x, wl, att, wij, bias, rowptr, src, prelu_slope):
        out = ops.gat_layer(rowptr, src, x, wl, att, wij, bias, prelu_slope=prelu_slope)
        ctx.save_for_backward(x, wl, att, wij, out, rowptr, src)
        ctx.prelu_slope = prelu_slope
        ctx.att_shape = att.shape
        return out

    @staticmethod
    def backward(ctx, gout):
        x, wl, att, wij, out, rowptr, src = ctx.saved_tensors
        gx, gwl, gatt, gwij, gbias = ops.gat_layer_bwd(rowptr, src, x, wl, att, wij, ctx.prelu_slope, out, gout.contiguous(),
                                                       need_gx=ctx.needs_input_grad[0])
        return gx, gwl, gatt.reshape(ctx.att_shape), gwij, gbias, None, None, None


class EdgeHead(torch.autograd.Function):
    """lin6(prelu(lin5([x_s || x_t]))) per edge (Teacher_model.py:54-59)."""

    @staticmethod
    def forward(ctx, x, w5, b5, w6, b6, src, dst, prelu_slope):
        pd = ops.edge_head(src, dst, x, w5, b5, prelu_slope, w6, b6)
        ctx.save_for_backward(x, w5, b5, w6, src, dst)
        ctx.prelu_slope = prelu_slope
        return pd

    @staticmethod
    def backward(ctx, gpd):
        x, w5, b5, w6, src, dst = ctx.saved_tensors
        gx, gw5, gb5, gw6, gb6 = ops.edge_head_bwd(src, dst, x, w5, b5, ctx.prelu_slope, w6, gpd.contiguous())
        return gx, gw5, gb5, gw6, gb6, None, None, None


_W2_STATUS = "1 = fewer predicted than target points, 2 = too many points for one problem (4096), 3 = NaN / Inf coordinates"


class DiagramLoss(torch.autograd.Function):
    """`wasserstein_distance(PD_hat, PD, order=p, enable_autodiff=True, num_models=1)` (wasserstein.py:198-379) of one or more
    (predicted, target) pairs: -> (loss [B], wxy [B], wxd [B]); only `loss` carries a gradient, like the reference's
    (wxy / wxd are the logged parts).

    infer=True: `wasserstein_distance_inference` (wasserstein.py:93-195, both diagrams may use the diagonal)
    -> (loss, wxy, wxd, wyd)."""

    @staticmethod
    def forward(ctx, pd_hat, xoff, target, yoff, order, infer):
        if infer:
            r = ops.w2_inference_matching(xoff, pd_hat.detach(), yoff, target, order=order, want_grad=True)
        else:
            r = ops.w2_partial_matching(xoff, pd_hat.detach(), yoff, target, order=order, want_grad=True)
        bad = r["status"] != 0
        if bool(bad.any()):
            raise ValueError("diagram loss: status %s (%s)" % (r["status"].tolist(), _W2_STATUS))
        ctx.save_for_backward(r["grad"], xoff)
        dt = pd_hat.dtype
        ctx.dtype = dt
        # (the tensors RETURNED are marked: a cast makes new ones, and a mark on the float64 originals would be lost)
        parts = [r[k].to(dt) for k in (("wxy", "wxd", "wyd") if infer else ("wxy", "wxd"))]
        ctx.mark_non_differentiable(*parts)
        return (r["loss"].to(dt),) + tuple(parts)

    @staticmethod
    def backward(ctx, gloss, *_unused):
        grad, xoff = ctx.saved_tensors
        cnt = xoff[1:] - xoff[:-1]
        per_point = torch.repeat_interleave(gloss.to(torch.float64), cnt)          # d total / d loss[b] for each predicted point
        return (grad * per_point.unsqueeze(1)).to(ctx.dtype), None, None, None, None, None


class DiagramImage(torch.autograd.Function):
    """The differentiable imager of Teacher_Model.forward(grad_PI=True) (Teacher_model.py:80-81 -> pimg.py:354-400): images of one
    or more predicted diagrams, [B, res*res] in the diagrams' dtype.  The reference detaches the coordinates inside the two
    normal-CDF factors (:392,395), so the gradient reaches a point through its weight only (`tlc_pi_raster_wgrad`)."""

    @staticmethod
    def forward(ctx, pd_hat, offs, res):
        pts = pd_hat.detach().to(torch.float64).contiguous()
        ctx.save_for_backward(pts, offs)
        ctx.res, ctx.dtype = res, pd_hat.dtype
        return engine.pi_raster(offs, pts, res).to(pd_hat.dtype)

    @staticmethod
    def backward(ctx, gimg):
        pts, offs = ctx.saved_tensors
        g = engine.pi_raster_wgrad(offs, pts, gimg.to(torch.float64), ctx.res)
        return g.to(ctx.dtype), None, None


def diagram_image(pd_hat, offs=None, res=5):
    if offs is None:
        offs = torch.tensor([0, pd_hat.shape[0]], dtype=torch.int64, device=pd_hat.device)
    return DiagramImage.apply(pd_hat, offs, int(res))


def gat_layer(x, wl, att, wij, bias, rowptr, src, prelu_slope=-1.0):
    return GatLayer.apply(x, wl, att, wij, bias, rowptr, src, float(prelu_slope))


def edge_head(x, w5, b5, w6, b6, src, dst, prelu_slope):
    return EdgeHead.apply(x, w5, b5, w6, b6, src, dst, float(prelu_slope))


def diagram_loss(pd_hat, target, order=2, xoff=None, yoff=None, infer=False):
    """-> (loss, wxy, wxd) per problem; infer=True: (loss, wxy, wxd, wyd) of the evaluation distance."""
    dev = pd_hat.device
    if xoff is None:
        xoff = torch.tensor([0, pd_hat.shape[0]], dtype=torch.int64, device=dev)
    if yoff is None:
        yoff = torch.tensor([0, target.shape[0]], dtype=torch.int64, device=dev)
    return DiagramLoss.apply(pd_hat, xoff, target, yoff, int(order), bool(infer))
