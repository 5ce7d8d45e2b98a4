"""tf.get_variable-style lazy variable store on the GPU (variable names = the reference's TF names)."""
import math

import torch


class VariableStore(torch.nn.Module):
    """Creates fp32 nn.Parameters on first use with the reference's initialisers and keeps them under their TF
    names ('conv1/conv1_1/conv2d/kernel' HWIO, '.../bias', 'l2_norm_layer_3/weight', ...)."""

    def __init__(self, device="cuda", seed=20180817):
        super().__init__()
        self.device = torch.device(device)
        self.gen = torch.Generator(device="cpu").manual_seed(seed)
        self.vars = torch.nn.ParameterDict()
        self.order = []
        self.fuse_groups = []       # [(names, axis)]: variables a layer consumes concatenated along `axis` (loc_i | cls_i heads)
        self.fused = {}             # names tuple -> concatenated tensor living in the trainer's flat buffer (FlatParams)

    @staticmethod
    def _key(name):
        return name.replace(".", "|")

    def get(self, name, shape, init):
        k = self._key(name)
        if k not in self.vars:
            shape = tuple(int(s) for s in shape)
            if init == "glorot":                       # tf.glorot_uniform_initializer on an HWIO kernel (net/sfd_net.py:65)
                kh, kw, ci, co = shape
                lim = math.sqrt(6.0 / (kh * kw * ci + kh * kw * co))
                v = (torch.rand(shape, generator=self.gen) * 2.0 - 1.0) * lim
            elif init == "glorot_oihw":                # deform kernel (Cout,Cin,kh,kw): fans from the shape as given
                co, ci, kh, kw = shape
                lim = math.sqrt(6.0 / (co * ci * kh + co * ci * kw))
                v = (torch.rand(shape, generator=self.gen) * 2.0 - 1.0) * lim
            elif init == "zeros":
                v = torch.zeros(shape)
            else:
                v = torch.full(shape, float(init))
            self.vars[k] = torch.nn.Parameter(v.to(self.device))
            self.order.append(name)
        p = self.vars[k]
        assert tuple(p.shape) == tuple(shape), (name, tuple(p.shape), tuple(shape))
        return p

    def fuse(self, names, axis=-1):
        """Declares that `names` are consumed concatenated along `axis` (one conv for loc_i | cls_i).  Returns the concatenated
        tensor once FlatParams has laid the members out as strided views of one block (no cat / split / re-packing per step), else
        None and the caller concatenates.  axis = "blockdiag" (two HWIO kernels): the members sit on the diagonal of one
        [kh, kw, c1 + c2, o1 + o2] kernel over the channel concatenation of their two inputs (zeros elsewhere).  axis = "plus"
        (a k x 1 and a 1 x k kernel of the same input): one k x k kernel with the members in its middle column / middle row and their
        outputs side by side.  axis = "hwio" (ONE OIHW kernel, the deformable convolution's): stored as the HWIO GEMM operand, the
        variable being the permuted view."""
        key = tuple(names)
        if all(key != k for k, _ in self.fuse_groups):
            self.fuse_groups.append((key, axis))
        t = self.fused.get(key)
        if t is None and not torch.is_grad_enabled() and all(self._key(k) in self.vars for k in key):
            t = self._inference_block(key, axis)
        return t

    def build(self, names, axis):
        """The fused tensor of `names` built with differentiable torch ops (plain autograd without a trainer: its gradient is sliced apart
        again by autograd; zeros' places receive nothing)."""
        ps = [self.vars[self._key(k)] for k in names]
        if axis == "blockdiag":
            (p1, p2) = ps
            z1 = p1.new_zeros(tuple(p1.shape[:3]) + (p2.shape[3],))
            z2 = p2.new_zeros(tuple(p2.shape[:3]) + (p1.shape[3],))
            return torch.cat([torch.cat([p1, z1], dim=3), torch.cat([z2, p2], dim=3)], dim=2).contiguous()
        if axis == "plus":
            (p1, p2) = ps                            # [k, 1, c, o1] in the middle column, [1, k, c, o2] in the middle row
            h = p1.shape[0] // 2
            pad = torch.nn.functional.pad
            return torch.cat([pad(p1, (0, 0, 0, 0, h, h)), pad(p2, (0, 0, 0, 0, 0, 0, h, h))], dim=3).contiguous()
        if axis == "hwio":                           # one OIHW kernel as the [1, 1, kh * kw * C, Cout] GEMM operand (k = tap * C + c)
            (p1,) = ps
            co, ci, kh, kw = p1.shape
            return p1.permute(2, 3, 1, 0).reshape(1, 1, kh * kw * ci, co).contiguous()
        return torch.cat(ps, dim=axis).contiguous()

    def _inference_block(self, key, axis):
        """No trainer has laid the block out and no gradient is being tracked (evaluation scripts): the fused tensor is built once and kept
        while its members are unchanged; `_danhip_grad = None` marks it for ops' packed-weight cache (packed once, not per call)."""
        ps = [self.vars[self._key(k)] for k in key]
        # What can change the members: an in-place torch op on the Parameter (its version counter), a re-pointed storage (FlatParams),
        # and writers that neither bumps — `p.data.copy_()` (`.data` carries its own counter) and the optimizer / checkpoint kernels that
        # write through raw pointers.  Those all advance ops.WEIGHT_EPOCH (FlatParams.sgd_step, checkpoint._bump_weight_epoch,
        # load_tf_named), so the epoch is part of the stamp (a restore after a gradient-free forward kept the old block).
        from .. import ops
        stamp = (ops.WEIGHT_EPOCH,) + tuple((p._version, p.data_ptr()) for p in ps)
        cache = self.__dict__.setdefault("_infer_blocks", {})
        hit = cache.get((key, axis))
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            t = self.build(key, axis)
        t._danhip_grad = None
        if axis == "blockdiag":
            low = t[:, :, ps[0].shape[2]:, :]
            low._danhip_grad = None
            t._danhip_lower = low
        cache[(key, axis)] = (stamp, t)
        return t

    def buffer(self, name, shape, init):
        """Non-trainable state (batch-norm moving averages): a plain device tensor kept under its TF name."""
        if not hasattr(self, "bufs"):
            self.bufs = {}
        if name not in self.bufs:
            self.bufs[name] = torch.full(tuple(int(s) for s in shape), float(init), dtype=torch.float32, device=self.device)
        return self.bufs[name]

    def named(self):
        return [(n, self.vars[self._key(n)]) for n in self.order]

    def load_tf_named(self, tensors):
        """tensors: {tf_name: array/tensor} (e.g. from the CPU oracle or a converted checkpoint)."""
        for n, t in tensors.items():
            t = torch.as_tensor(t, dtype=torch.float32)
            k = self._key(n)
            if k in self.vars:
                with torch.no_grad():
                    self.vars[k].copy_(t.to(self.device))
                from .. import ops                    # members of a fused block are views: cached blocks / 16-bit packings key on the epoch
                ops.WEIGHT_EPOCH += 1
            else:
                self.vars[k] = torch.nn.Parameter(t.to(self.device).clone())
                self.order.append(n)

    def export_tf_named(self):
        return {n: p.detach().cpu().clone() for n, p in self.named()}
