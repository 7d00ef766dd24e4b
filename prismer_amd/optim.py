"""Drop-in optimizer for the reference loop: `torch.optim.AdamW`'s interface (train_caption.py:111-112:
`torch.optim.AdamW(params=filter(lambda p: p.requires_grad, model.parameters()), lr=..., weight_decay=...)`) on top of the library's fused
AdamW kernel.

The drop-in modules keep their parameters as views of one flat fp32 master buffer per top module (prismer_amd/store.py) and hand autograd
gradients that are views of one flat gradient buffer in the same layout.  `torch.optim.AdamW` walks them with seven multi-tensor passes
(4.9 ms per step for Prismer-BASE's 242 M trainable parameters, rocprofv3) and leaves the bf16 shadows stale (a 0.4-ms cast on the next
forward); this class recognises a store whose trainable parameters are ALL in the step and whose gradients alias one buffer in layout order, and
issues ONE `ph_adamw` launch for it (1.3 ms for both stores, shadows refreshed in the same pass).  Everything else -- foreign parameters, a store
only partly in the optimizer, gradients that were re-assigned -- takes a plain per-tensor AdamW with identical arithmetic, so the class is safe
as a general replacement.  Same update rule as torch (decoupled weight decay, bias corrections; amsgrad / maximize are not offered)."""
import math
import weakref

import torch

from . import ops

_STORE_OF = {}        # id(Parameter) -> (weakref(Parameter), weakref(ParamStore), name)      (filled by ParamStore.attach; keyed by identity: Tensor.__eq__ is elementwise)


def register(store):
    for n, p in store.params.items():
        k = id(p)
        _STORE_OF[k] = (weakref.ref(p, lambda _, k=k: _STORE_OF.pop(k, None)), weakref.ref(store), n)


def _owner(p):
    ent = _STORE_OF.get(id(p))
    if ent is None or ent[0]() is not p:
        return None, None
    return ent[1](), ent[2]


class AdamW(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1 and 0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError('invalid AdamW hyper-parameter')
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self._flat = {}                      # id(store) -> dict(m, v, step, hyper)
        self.fused_launches = 0              # (tests / bench: store-wide launches / per-tensor updates the last step() issued)
        self.plain_updates = 0

    # ---------------------------------------------------------------------------------------------------------------- fused path
    @staticmethod
    def _aliased_flat_grad(st, names):
        """the flat fp32 gradient tensor [n_train] the gradients of `names` are views of (store layout), or None"""
        first = st.params[names[0]].grad
        if first is None or first.dtype != torch.float32:
            return None
        base = first.data_ptr() - 4 * st.offset[names[0]]
        stor = first.untyped_storage()
        for n in names:
            g = st.params[n].grad
            if g is None or g.dtype != torch.float32 or not g.is_contiguous() or g.data_ptr() != base + 4 * st.offset[n] or \
                    g.untyped_storage().data_ptr() != stor.data_ptr():
                return None
        off = (base - stor.data_ptr()) // 4
        if base < stor.data_ptr() or (off + st.n_train) * 4 > stor.nbytes():
            return None
        return torch.empty(0, dtype=torch.float32, device=first.device).set_(stor, off, (st.n_train,), (1,))

    def _store_state(self, st):
        s = self._flat.get(id(st))
        if s is None:
            dev = st.master.device
            s = self._flat[id(st)] = dict(m=torch.zeros(st.n_train, dtype=torch.float32, device=dev), v=torch.zeros(st.n_train, dtype=torch.float32, device=dev),
                                          step=0, step_t=torch.tensor(0.0), hyper=torch.zeros(3, dtype=torch.float32, device=dev), store=weakref.ref(st))
            for n in st.names:                                   # per-parameter views (and ONE shared step counter): Optimizer.state_dict() keeps its usual shape
                if n in st.trainable:
                    o, k = st.offset[n], st.numel[n]
                    old = self.state.get(st.params[n])           # (state that load_state_dict() put there before the first fused step)
                    if old:
                        s['m'][o:o + k].copy_(old['exp_avg'].reshape(-1)); s['v'][o:o + k].copy_(old['exp_avg_sq'].reshape(-1))
                        s['step'] = int(old['step'].item()) if torch.is_tensor(old['step']) else int(old['step'])
                    self.state[st.params[n]] = dict(step=s['step_t'], exp_avg=s['m'][o:o + k].view(st.shape[n]), exp_avg_sq=s['v'][o:o + k].view(st.shape[n]))
            s['step_t'].fill_(float(s['step']))
        return s

    def _steps_agree(self, st, names):
        """per-tensor state (loaded, or from earlier per-tensor steps) can move into the flat buffers only if every tensor took the same number of steps"""
        seen = {float(self.state[st.params[n]]['step']) if self.state.get(st.params[n]) else 0.0 for n in names}
        return len(seen) == 1

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        self.fused_launches = self.plain_updates = 0
        for group in self.param_groups:
            lr, (b1, b2), eps, wd = group['lr'], group['betas'], group['eps'], group['weight_decay']
            by_store, plain = {}, []
            for p in group['params']:
                if p.grad is None:
                    continue
                st, name = _owner(p)
                if st is not None and st.attached and st.params.get(name) is p:
                    by_store.setdefault(id(st), (st, []))[1].append(name)
                else:
                    plain.append(p)
            for st, names in by_store.values():
                have = set(names)
                names = [n for n in st.names if n in have]                       # layout order
                flat = self._aliased_flat_grad(st, names) if have == st.trainable else None
                s = self._flat.get(id(st))
                if flat is None or (s is None and not self._steps_agree(st, names)):
                    if s is not None:                                            # leaving the fused path: every tensor gets its own step counter back
                        for n in st.names:
                            if self.state.get(st.params[n]):
                                self.state[st.params[n]]['step'] = s['step_t'].clone()
                        del self._flat[id(st)]
                    plain += [st.params[n] for n in names]                       # (partly covered store / foreign gradients / uneven per-tensor state)
                    continue
                s = self._store_state(st)
                if s['m'].numel() != st.n_train:
                    raise RuntimeError('the parameter store was re-laid out (requires_grad flags changed) after this optimizer took its first step: build a new optimizer')
                s['step'] += 1
                t = s['step']
                ops.store_words(s['hyper'], (lr, 1.0 - b1 ** t, 1.0 - b2 ** t))      # (kernel arguments: no host-to-device copy, no stream wait)
                ops.adamw(st.master, flat, s['m'], s['v'], st.shadow, st.n_train, s['hyper'], b1, b2, eps, wd, 1.0, zero_grad=False)
                st.refresh_derived()                                             # (the launch wrote the bf16 shadows and bumped no Parameter._version: the store does not see itself as stale)
                s['step_t'] += 1
                self.fused_launches += 1
            self.plain_updates += len(plain)
            for p in plain:                                                      # torch.optim.AdamW's arithmetic, one tensor at a time
                stt = self.state[p]
                if not stt:
                    stt['step'] = torch.tensor(0.0)
                    stt['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    stt['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                stt['step'] += 1
                t = int(stt['step'].item())
                g = p.grad
                p.mul_(1.0 - lr * wd)
                stt['exp_avg'].lerp_(g, 1.0 - b1)
                stt['exp_avg_sq'].mul_(b2).addcmul_(g, g, value=1.0 - b2)
                denom = (stt['exp_avg_sq'].sqrt() / math.sqrt(1.0 - b2 ** t)).add_(eps)
                p.addcdiv_(stt['exp_avg'], denom, value=-lr / (1.0 - b1 ** t))
        return loss

    def load_state_dict(self, state_dict):
        """torch's loader replaces the state tensors by copies: fold them back into the flat moment buffers the fused launches use"""
        super().load_state_dict(state_dict)
        for s in self._flat.values():
            st = s['store']()
            if st is None:
                continue
            for n in st.names:
                if n in st.trainable and st.params[n] in self.state and self.state[st.params[n]]:
                    o, k = st.offset[n], st.numel[n]
                    ld = self.state[st.params[n]]
                    s['m'][o:o + k].copy_(ld['exp_avg'].reshape(-1)); s['v'][o:o + k].copy_(ld['exp_avg_sq'].reshape(-1))
                    s['step'] = int(ld['step'].item()) if torch.is_tensor(ld['step']) else int(ld['step'])
                    s['step_t'].fill_(float(s['step']))
                    ld['step'] = s['step_t']
                    ld['exp_avg'], ld['exp_avg_sq'] = s['m'][o:o + k].view(st.shape[n]), s['v'][o:o + k].view(st.shape[n])
