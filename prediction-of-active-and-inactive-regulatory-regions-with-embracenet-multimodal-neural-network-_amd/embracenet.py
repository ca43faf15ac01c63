"""EmbraceNet fusion module and the multimodal classifier built on it -- drop-in interface of
BIOINF_tesi/models/EmbraceNetMultimodal.py (EmbraceNet :12-90, EmbraceNetMultimodal :94-193),
executed by hand-written gfx950 kernels (csrc/).  Same class names, constructor/forward signatures,
parameter names and shapes (``embracenet.docking_{0,1}.{weight,bias}``, ``post.{3i}.*``), same
Optuna ``trial`` call order, same errors (AssertionError on a modality-count mismatch, RuntimeError on
an invalid selection distribution).

RNG
---
``rng_mode == "host"``  (parity mode): every random number is drawn from a CPU ``torch.Generator`` in
  exactly the reference's order -- rand(1) fp32, rand(B) fp32 (modality dropout, :179-181), then the
  B*c fp64 uniforms torch.multinomial consumes (:84) -- and injected into the kernel, which reproduces
  the CPU index tensor bit for bit.
``rng_mode == "philox"`` (default): the kernels draw from Philox4x32-10 keyed on (seed, step, global
  element index); no host round trip, no device->host sync, graph-capturable, invariant to how the batch
  is sharded across GPUs.
"""

import torch
import torch.nn as nn

from . import functional as F_
from ._lib import STATUS_INVALID_DISTRIBUTION
from .prenets import CNN_pre, FFNN_pre




class _RngMixin:
    def _init_rng(self):
        self.rng_mode = "philox"
        self.generator = None          # host mode: CPU generator (None = torch's default CPU generator)
        self.rng_seed = int(torch.initial_seed()) & 0xFFFFFFFFFFFFFFFF
        self.rng_row0 = 0              # global index of local row 0 (data parallel)
        self._step_dev = None          # uint64-as-int64 device counter, created lazily
        self.check_distribution = None  # None: check (sync) in host mode only

    def set_rng(self, mode, generator=None, seed=None, row0=None):
        if mode not in ("host", "philox"):
            raise ValueError("rng mode must be 'host' or 'philox'")
        self.rng_mode, self.generator = mode, generator
        if seed is not None:
            self.rng_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        if row0 is not None:
            self.rng_row0 = int(row0)
        return self

    def _rng_state(self, device):
        device = torch.device(device)
        if device.type == "cuda" and device.index is None:      # "cuda" and "cuda:0" name the same counter (as optim._Fused)
            device = torch.device("cuda", torch.cuda.current_device())
        if self._step_dev is None:
            self._step_dev = torch.zeros(1, dtype=torch.int64, device=device)
        elif self._step_dev.device != device:                   # a real device change: the step value moves along
            self._step_dev = self._step_dev.to(device)
        return F_.RngState(self.rng_seed, 0, self._step_dev, self.rng_row0)

    def _advance_step(self):
        from . import _lib
        _lib.check(_lib.lib().emb_counter_add(self._step_dev.data_ptr(), 1, _lib.stream()), "emb_counter_add")

    def __getstate__(self):
        st = self.__dict__.copy()
        for k in ("_step_dev", "generator", "last_code", "last_status", "_sel_dev", "_sel_key", "_status"):
            if k in st:
                st[k] = None
        return st


class EmbraceNet(nn.Module, _RngMixin):
    def __init__(self, device, input_size_list, embracement_size=256, bypass_docking=False):
        super().__init__()
        self.device = device
        self.input_size_list = input_size_list
        self.embracement_size = embracement_size
        self.bypass_docking = bypass_docking
        if not bypass_docking:
            for i, size in enumerate(input_size_list):
                setattr(self, "docking_%d" % i, nn.Linear(size, embracement_size))
        self.compute_dtype = None      # None: follow the parameters' dtype; torch.bfloat16: bf16 shadows
        self.last_code = None          # [B,c] uint8 of the latest forward (bit0 = selected modality)
        self._init_rng()

    def _prepare(self, B, dev, availabilities, selection_probabilities, _device_dropout):
        """Selection cdf of the batch (:63-76) -- independent of the docking inputs, so a caller may run it early / on
        another stream and hand the result to forward(_prep=...)."""
        rng = self._rng_state(dev)
        if selection_probabilities is None:                       # :70-71
            selection_probabilities = torch.ones(1, 2, dtype=torch.float32, device=dev)
        p = selection_probabilities.to(device=dev, dtype=torch.float32)
        if p.dim() == 2 and p.shape[0] not in (1, B):
            raise ValueError("selection_probabilities must be [B, M] or [M]")
        avail = None if availabilities is None else availabilities.to(device=dev, dtype=torch.float32)
        if getattr(self, "_status", None) is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)       # sticky bits, cleared when read
        return F_.select_prep(p, avail, B, rng=rng, device_dropout=_device_dropout, status=self._status)

    def _select_inline(self, B, dev, availabilities, selection_probabilities, _device_dropout):
        if selection_probabilities is None:                       # :70-71
            selection_probabilities = torch.ones(1, 2, dtype=torch.float32, device=dev)
        p = selection_probabilities.to(device=dev, dtype=torch.float32)
        if p.dim() == 2 and p.shape[0] not in (1, B):
            raise ValueError("selection_probabilities must be [B, M] or [M]")
        avail = None if availabilities is None else availabilities.to(device=dev, dtype=torch.float32)
        if getattr(self, "_status", None) is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)       # sticky bits, cleared when read
        return F_.SelectInline(p, avail, _device_dropout, self._status), self._status

    def forward(self, input_list, availabilities=None, selection_probabilities=None, _device_dropout=False,
                _advance=True, _prep=None):
        assert len(input_list) == len(self.input_size_list)
        if len(input_list) != 2:
            return self._forward_m(list(input_list), availabilities, selection_probabilities, _device_dropout, _advance)
        x0, x1 = input_list
        B, c = x0.shape[0], self.embracement_size
        dev = x0.device
        if self.bypass_docking:                                   # :54-55: the inputs are the docking outputs
            if tuple(x0.shape) != (B, c) or tuple(x1.shape) != (B, c):
                raise ValueError("bypass_docking: input data must have a shape of [batch_size, embracement_size]")
            T = self.compute_dtype or x0.dtype
        else:
            T = self.compute_dtype or self.docking_0.weight.dtype
        rng = self._rng_state(dev)
        check = self.check_distribution if self.check_distribution is not None else (self.rng_mode == "host")
        if _prep is None and self.rng_mode == "philox" and not check:
            # device RNG, no host-side validity check wanted: the row thresholds are computed inside the forward launch
            cdf0, status = self._select_inline(B, dev, availabilities, selection_probabilities, _device_dropout)
        else:
            cdf0, status = _prep if _prep is not None else self._prepare(B, dev, availabilities, selection_probabilities,
                                                                         _device_dropout)

        u = None
        if self.rng_mode == "host":                               # replay of torch.multinomial's draws (:84)
            u = torch.rand(B * c, dtype=torch.float64, generator=self.generator).view(B, c).to(dev, non_blocking=True)
        if check and int(status.item()) & STATUS_INVALID_DISTRIBUTION:
            status.zero_()
            raise RuntimeError("invalid multinomial distribution (encountering probability entry < 0)")
        self.last_status = status

        if self.bypass_docking:
            E, code = F_.embrace_bypass(x0, x1, cdf0, u=u, rng=rng, compute_dtype=T)
        else:
            E, code = F_.embrace(x0, x1, self.docking_0.weight, self.docking_0.bias, self.docking_1.weight,
                                 self.docking_1.bias, cdf0, u=u, rng=rng, compute_dtype=T)
        self.last_code = code
        self._code_is_index = False
        if _advance:
            self._advance_step()
        return E

    def _forward_m(self, xs, availabilities, selection_probabilities, _device_dropout, _advance):
        """len(input_list) != 2 (:46-48; no call site of the reference): docking layers as M Linear+ReLU launches (:52-60),
        then the selection pass over their outputs (:63-88)."""
        M, c = len(xs), self.embracement_size
        if not 1 <= M <= 8:
            raise NotImplementedError("the selection kernels take 1..8 modalities")
        if _device_dropout:
            raise NotImplementedError("device-side modality dropout is the two-modality model's (:178-182)")
        B, dev = xs[0].shape[0], xs[0].device
        rng = self._rng_state(dev)
        if self.bypass_docking:                                   # :54-55
            if any(tuple(x.shape) != (B, c) for x in xs):
                raise ValueError("bypass_docking: input data must have a shape of [batch_size, embracement_size]")
            T = self.compute_dtype or xs[0].dtype
            D = xs
        else:
            T = self.compute_dtype or self.docking_0.weight.dtype
            D = [F_.linear(x, getattr(self, "docking_%d" % m).weight, getattr(self, "docking_%d" % m).bias, relu=True,
                           rng=rng, compute_dtype=T) for m, x in enumerate(xs)]
        p = selection_probabilities
        p = torch.ones(1, M, dtype=torch.float32, device=dev) if p is None else p.to(device=dev, dtype=torch.float32)   # :70-71
        avail = None if availabilities is None else availabilities.to(device=dev, dtype=torch.float32)
        if getattr(self, "_status", None) is None or self._status.device != dev:
            self._status = torch.zeros(1, dtype=torch.int32, device=dev)
        cdf, status = F_.select_prep_m(p, avail, B, M, status=self._status)
        u = None
        if self.rng_mode == "host":                               # replay of torch.multinomial's draws (:84)
            u = torch.rand(B * c, dtype=torch.float64, generator=self.generator).view(B, c).to(dev, non_blocking=True)
        check = self.check_distribution if self.check_distribution is not None else (self.rng_mode == "host")
        if check and int(status.item()) & STATUS_INVALID_DISTRIBUTION:
            status.zero_()
            raise RuntimeError("invalid multinomial distribution (encountering probability entry < 0)")
        self.last_status = status
        E, code = F_.embrace_select(D, cdf, u=u, rng=rng, compute_dtype=T)
        self.last_code = code
        self._code_is_index = True
        if _advance:
            self._advance_step()
        return E

    def modality_indices(self):
        """[B, c] int64 index tensor of the latest forward (what torch.multinomial returned in the reference)."""
        if getattr(self, "_code_is_index", False):                # M != 2: the code byte is the index itself
            return self.last_code.to(torch.int64)
        return (self.last_code & 1).to(torch.int64)


class EmbraceNetMultimodal(nn.Module, _RngMixin):
    def __init__(self, trial, cell_line, task, device, in_features_FFNN, n_classes=2, args=None,
                 embracenet_dropout=True):
        super().__init__()
        self.trial = trial
        self.cell_line = cell_line
        self.device = device
        self.n_classes = n_classes
        self.embracenet_dropout = embracenet_dropout
        self.args = args

        self.FFNN = FFNN_pre(self.trial, in_features_FFNN, device=self.device)
        self.CNN = CNN_pre(self.trial, device=self.device)
        self.FFNN_pre_output_size = self.FFNN.output_size
        self.CNN_pre_output_size = self.CNN.output_size

        embracement_size = self.trial.suggest_categorical("EMBRACENET_embracement_size", [512, 768, 1024])
        self.embracenet = EmbraceNet(device=self.device,
                                     input_size_list=[self.FFNN_pre_output_size, self.CNN_pre_output_size],
                                     embracement_size=embracement_size)

        n_post_layers = self.trial.suggest_int("n_post_layers", 0, 2)
        widths = ([32, 64, 128, 256, 512], [16, 32, 64, 128, 256])
        stack, width = [], embracement_size
        for i in range(n_post_layers):
            out = self.trial.suggest_categorical(f"EMBRACENET_n_units_l{i}", widths[i])
            p = self.trial.suggest_categorical(f"EMBRACENET_dropout_l{i}", [0.0, 0.2, 0.3, 0.5])
            stack += [nn.Linear(width, out), nn.ReLU(), nn.Dropout(p)]
            width = out
        stack.append(nn.Linear(width, self.n_classes))
        self.post = nn.Sequential(*stack)

        s = self.trial.suggest_float("selection_probabilities_FFNN", 0.0, 1.0)
        self.selection_probabilities = torch.tensor([s, 1.0 - s])     # plain attribute, not in the state dict (:157)
        self.compute_dtype = None
        self._init_rng()

    # -- configuration ---------------------------------------------------------------------------
    def set_compute_dtype(self, dtype):
        """None: kernels run in the parameters' dtype (fp32 / fp64).  torch.bfloat16: bf16 activations and
        weight shadows, fp32 master parameters, fp32 accumulation."""
        self.compute_dtype = dtype
        self.embracenet.compute_dtype = dtype
        return self

    def set_rng(self, mode, generator=None, seed=None, row0=None):
        _RngMixin.set_rng(self, mode, generator, seed, row0)
        self.embracenet.set_rng(mode, generator, self.rng_seed, self.rng_row0)
        return self

    # -- forward ------------------------------------------------------------------------------------
    def _post_forward(self, y, rng, T):
        mods = list(self.post)
        i, layer_id, layers = 0, 0, []
        while i < len(mods):
            lin = mods[i]
            relu = i + 1 < len(mods) and isinstance(mods[i + 1], nn.ReLU)
            p = 0.0
            if relu and i + 2 < len(mods) and isinstance(mods[i + 2], nn.Dropout):
                p = float(mods[i + 2].p) if self.training else 0.0
            layers.append((lin.weight, lin.bias, relu, p, layer_id))
            i += 3 if relu else 1
            layer_id += 1
        arm, self._fused_loss = getattr(self, "_fused_loss", None), None      # armed for ONE forward
        if arm is not None:
            if not self.fused_loss_ready(y.shape[0]):
                raise RuntimeError("arm_fused_loss: check fused_loss_ready() first")
            if len(layers) > 1:                                  # hidden post layers, then the head takes the loss with it
                y = F_.mlp(y, layers[:-1], rng=rng, compute_dtype=T)
            w, b = layers[-1][:2]
            # no hidden post layer: the head reads the fusion layer's output itself and prepares that layer's backward
            code = self.embracenet.last_code if (len(layers) == 1 and getattr(self, "premask_in_head", True)) else None
            return F_.head_ce(y, w, b, arm, compute_dtype=T, code=code)
        return F_.mlp(y, layers, rng=rng, compute_dtype=T)

    def fused_loss_ready(self, B):
        """True when the final Linear(width, 2), the loss and the head's backward can run as ONE launch (csrc/head.hip):
        fp32 / bf16 compute on fp32 masters.  A trainer then calls arm_fused_loss(functional.FusedLoss(...)) right before
        the forward and ``logits.backward(<any tensor of that shape>)`` after it, instead of taking the loss itself."""
        last = [m for m in self.post if isinstance(m, nn.Linear)][-1]
        T = self.compute_dtype or last.weight.dtype
        return last.weight.is_cuda and last.weight.dtype == torch.float32 and last.out_features == 2 and \
            F_.head_ce_supported(B, last.in_features, T)

    def arm_fused_loss(self, fused_loss):
        self._fused_loss = fused_loss

    def forward(self, x, availabilities=None, selection_probabilities=None, is_training=False,
                embracenet_dropout=True):
        x_FFNN, x_CNN = x
        dev = x_FFNN.device
        T = self.compute_dtype or self.embracenet.docking_0.weight.dtype
        self.embracenet.rng_mode, self.embracenet.generator = self.rng_mode, self.generator
        self.embracenet.rng_seed, self.embracenet.rng_row0 = self.rng_seed, self.rng_row0
        rng = self.embracenet._rng_state(dev) if x_FFNN.is_cuda else None
        self.FFNN.compute_dtype = self.CNN.compute_dtype = self.compute_dtype
        B = x_FFNN.shape[0]
        device_dropout = False
        if is_training and embracenet_dropout:                              # :178-182
            if self.rng_mode == "host":
                r = torch.rand(1, generator=self.generator)[0]
                if r >= 0.5:
                    t = torch.round(torch.rand([B], generator=self.generator)).to(torch.int64)
                    availabilities = nn.functional.one_hot(t, num_classes=2).float()
            else:
                device_dropout = True
        # the reference ignores the selection_probabilities argument and always uses its own (:184-187)
        sp = self.selection_probabilities
        key = (sp.data_ptr(), sp._version, dev)
        if getattr(self, "_sel_key", None) != key:      # device copy made once, not per step (no H2D in a graph)
            self._sel_dev, self._sel_key = sp.detach().to(device=dev, dtype=torch.float32).view(1, 2), key
        p = self._sel_dev
        # The two pre-networks are independent chains.  The epigenomic MLP is tiny and latency-bound, so its launches RIDE on
        # kernels of the sequence CNN (csrc/rider.h): the forward is parked here and carried by the CNN's first kernel; its
        # autograd node is attached after the CNN's, so its backward runs first and is carried by the CNN's BatchNorm-backward
        # pass.  (ride_prenets = False, or a stack the fused kernels do not take: plain launches.  A side stream for the MLP
        # was measured slower than one stream: the fork / join edges of the captured graph cost more than they hide.)
        handle = None
        if rng is not None and getattr(self, "ride_prenets", True) and T == torch.bfloat16:
            handle = self.FFNN.prelaunch(x_FFNN, rng=rng)
        if handle is not None:
            h1 = self.CNN(x_CNN, rng=rng)
            h0 = self.FFNN.attach(handle)
        else:
            h0, h1 = self.FFNN(x_FFNN, rng=rng), self.CNN(x_CNN, rng=rng)
        E = self.embracenet([h0, h1], availabilities=availabilities, selection_probabilities=p,
                            _device_dropout=device_dropout, _advance=False)
        out = self._post_forward(E, rng, T)
        if not getattr(self, "defer_step_tick", False):     # a trainer may fold the tick into its loss kernel (step_counter)
            self.embracenet._advance_step()
        return out

    def step_counter(self, device):
        """int64 device scalar added to the RNG step of every kernel of a forward.  It must advance by one after each
        forward: `forward` does that with one tiny launch unless ``defer_step_tick`` is set, in which case the caller
        hands this tensor to functional.weighted_ce_with_grad(ticks=...) (training.StepRunner does)."""
        self.embracenet._rng_state(device)
        return self.embracenet._step_dev
