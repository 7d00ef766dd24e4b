"""Flat parameter storage for the HIP layer programs.

The nn.Module shells (prismer_amd/modules) own ordinary fp32 nn.Parameters with the reference's names, so
state_dict()/load_state_dict()/optimizers/DDP keep working.  `ParamStore.attach()` re-homes their storage into
flat buffers sized for HBM-resident training:

    master  fp32  [n_train | n_frozen]   <- Parameter.data are views into this
    shadow  bf16  same layout            <- what the MFMA kernels read (refreshed by the fused AdamW / refresh())
    grad    fp32  [n_train]              <- weight-gradient GEMMs accumulate straight into views of this

Trainable parameters come first, so one fused AdamW launch and one bucketed all-reduce sweep cover them.
Decoder q/k/v projections are laid out back-to-back ([3H,H] weight, [3H] bias) so that the layer programs issue
ONE packed QKV GEMM although the reference keeps three nn.Linear (roberta.py:86-92); likewise cross-attention k/v.
"""
import re
from collections import OrderedDict

import torch

from . import ops

ALIGN = 64   # elements: keeps every view 128-B (bf16) / 256-B (fp32) aligned


_CROSS_KEY = re.compile(r'(.*\.layer)\.(\d+)\.1\.self\.key\.weight$')


def _reorder_qkv(names):
    """[..self.query.weight, ..self.query.bias, ..self.key.weight, ..] -> weights (q,k,v) then biases (q,k,v).
    Cross-attention K/V projections (encoder.layer.{l}.1.self.{key,value}) of ALL layers are additionally placed
    back-to-back -- [k0,v0,k1,v1,...] weights, then the same for biases: they all project the same encoder output, so
    the decoder program runs them (and their dgrad / wgrad) as ONE GEMM of N = 2*H*layers (roberta.py:88-92 x 12)."""
    out, used = [], set()
    cross = [m for m in (_CROSS_KEY.match(n) for n in names) if m]
    if cross:
        pre = cross[0].group(1)
        layers = sorted(int(m.group(2)) for m in cross)
        grp = [f'{pre}.{l}.1.self.{w}.{s}' for s in ('weight', 'bias') for l in layers for w in ('key', 'value')]
        if all(g in names for g in grp):
            cross_grp, cross_at = grp, f'{pre}.{layers[0]}.1.self.key.weight'
            used.update(grp)
        else:
            cross_grp, cross_at = None, None
    else:
        cross_grp, cross_at = None, None
    for n in names:
        if n == cross_at:
            out += cross_grp
            continue
        if n in used:
            continue
        m = re.match(r'(.*\.self)\.query\.weight$', n)
        if m:
            p = m.group(1)
            grp = [f'{p}.{w}.{s}' for s in ('weight', 'bias') for w in ('query', 'key', 'value')]
            if all(g in names and g not in used for g in grp):
                out += grp
                used.update(grp)
                continue
        out.append(n)
        used.add(n)
    return out


class ParamStore:
    def __init__(self, module, prefix=''):
        self.module = module
        self.prefix = prefix
        self.attached = False
        self.native_grads = False      # True: backward leaves grads in self.grad only (Trainer); False: also returns them to autograd
        self.derived = {}              # name -> (bf16 tensor, refresh fn)
        self._grad_cur = None

    # ------------------------------------------------------------------------------------------ layout
    def attach(self):
        named = OrderedDict(self.module.named_parameters())          # duplicates (tied weights) removed by torch
        order = _reorder_qkv(list(named))
        train = [n for n in order if named[n].requires_grad]
        frozen = [n for n in order if not named[n].requires_grad]
        dev = next(iter(named.values())).device
        assert dev.type == 'cuda', 'ParamStore.attach() needs the module on the GPU (HIP path only)'
        self.names = train + frozen
        self.offset, self.numel, self.shape = {}, {}, {}
        off = 0
        for n in self.names:
            if n == (frozen[0] if frozen else None):
                off = (off + ALIGN - 1) // ALIGN * ALIGN
                self.n_train = off
            self.offset[n] = off
            self.numel[n] = named[n].numel()
            self.shape[n] = tuple(named[n].shape)
            off += (named[n].numel() + ALIGN - 1) // ALIGN * ALIGN
        if not frozen:
            self.n_train = off
        self.n_total = off
        self.trainable = set(train)
        self.master = torch.zeros(self.n_total, dtype=torch.float32, device=dev)
        self.shadow = torch.zeros(self.n_total, dtype=torch.bfloat16, device=dev)
        self.grad = torch.zeros(max(self.n_train, 1), dtype=torch.float32, device=dev)
        self.params = named
        with torch.no_grad():
            for n in self.names:
                v = self.master[self.offset[n]:self.offset[n] + self.numel[n]].view(self.shape[n])
                v.copy_(named[n].data)
                named[n].data = v
        self.attached = True
        self._freeze_sig = tuple(named[n].requires_grad for n in self.names)
        self.refresh()
        from . import optim
        optim.register(self)                 # (prismer_amd.optim.AdamW recognises these parameters and updates the store with one fused launch)
        return self

    def check_layout(self):
        """re-attach if requires_grad flags changed (Prismer.prepare_to_train may be called later)."""
        sig = tuple(self.params[n].requires_grad for n in self.names)
        if sig != self._freeze_sig:
            self.attach()

    # ------------------------------------------------------------------------------------------ views
    def f(self, name):
        """fp32 master view."""
        return self.params[name].data

    def w(self, name):
        """bf16 shadow view with the parameter's shape."""
        o = self.offset[name]
        return self.shadow[o:o + self.numel[name]].view(self.shape[name])

    def w2(self, name, rows, cols):
        """bf16 shadow of `rows*cols` elements starting at `name` (packed q|k|v style views)."""
        o = self.offset[name]
        return self.shadow[o:o + rows * cols].view(rows, cols)

    def fvec(self, name, n):
        o = self.offset[name]
        return self.master[o:o + n]

    def is_trainable(self, name):
        return name in self.trainable

    def g(self, name):
        """fp32 gradient view (None when frozen)."""
        if name not in self.trainable:
            return None
        o = self.offset[name]
        return self._grad_cur[o:o + self.numel[name]].view(self.shape[name])

    def g2(self, name, rows, cols):
        if name not in self.trainable:
            return None
        o = self.offset[name]
        return self._grad_cur[o:o + rows * cols].view(rows, cols)

    def gvec(self, name, n):
        if name not in self.trainable:
            return None
        o = self.offset[name]
        return self._grad_cur[o:o + n]

    # ------------------------------------------------------------------------------------------ per step
    def refresh(self):
        """bf16 shadows <- fp32 masters (after load_state_dict or a foreign optimizer step)."""
        ops.cast_to_bf16(self.master, self.shadow)
        self.refresh_derived()
        self._versions = {n: self.params[n]._version for n in self.names}

    def refresh_if_stale(self):
        if any(self.params[n]._version != v for n, v in self._versions.items()):
            self.refresh()

    def refresh_derived(self):
        """re-derive every buffer that is a function of the parameters (after an optimizer step).  Conv weight shadows
        (im2col column order) go through ONE grouped launch per 32 layers instead of a launch per layer."""
        conv, dgrad = [], []
        for t, fn, meta in self.derived.values():
            if meta is not None and meta[0] == 'dgrad':          # weight operand of the implicit data gradient (ops.conv_dgrad_shadows)
                _, name, Co, Ci, stride = meta
                dgrad.append((self.f(name), t, Co, Ci, stride))
            elif meta is not None:
                name, Co, Ci, ks, Kp = meta
                conv.append((self.f(name), t, Co, Ci, ks, Kp))
            else:
                fn(t)
        if conv:
            ops.conv_layout_grouped(conv, True)
        if dgrad:
            ops.conv_dgrad_shadows(dgrad)

    def derived_buffer(self, key, shape, fn, conv=None):
        if key not in self.derived:
            t = torch.zeros(shape, dtype=torch.bfloat16, device=self.master.device)
            fn(t)
            self.derived[key] = (t, fn, conv)
        return self.derived[key][0]

    def begin_grads(self):
        """gradient buffer for the coming backward: the persistent flat buffer in native mode (the Trainer zeroes
        it), a fresh zero buffer otherwise (its views are handed to autograd and become Parameter.grad)."""
        if self.native_grads:
            self._grad_cur = self.grad
        else:
            self._grad_cur = torch.zeros_like(self.grad)
        return self._grad_cur

    def grads_for_autograd(self, names):
        if self.native_grads:
            return [None] * len(names)
        return [self.g(n) for n in names]
