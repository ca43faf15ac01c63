"""Stock-PyTorch CPU restatement of the reference's EmbraceNetMultimodal train/eval step.

TEST INFRASTRUCTURE ONLY (see oracle/embrace_oracle.py header).  Used as
  * the full-model oracle for parity tests (pre-nets + embrace + post + loss + grads), and
  * ``bench.py``'s ``cpu_baseline`` leg (kind "port"), timed on the GPU box's host cores.
It is validated in the build container against the imported reference on identical
seeded inputs (tests/golden/make_golden.py asserts oracle == reference while writing the fixtures) and
pinned on the GPU box by the committed fixtures G2/G3/G9.

Reference citations (relative to /root/reference/BIOINF_tesi/models):
  FFNN_pre.py:18-49        epigenomic MLP
  CNN_pre.py:24-76         sequence CNN
  EmbraceNetMultimodal.py:34-90, :159-193   fusion + post stack
  utils/training_models_multimodal.py:132-163 (train step), :167-192 (eval step)
"""
import numpy as np
import torch
import torch.nn.functional as F


def conv_out_len(L, k):
    pad = (k - 1) // 2
    L = (L + 2 * pad - k) + 1
    return (L - 10) // 2 + 1


class OracleEmbraceNetMultimodal(torch.nn.Module):
    """hp: dict with the reference's Optuna parameter names (SURVEY 8b trial call order)."""

    def __init__(self, hp, in_features_FFNN, n_classes=2, dtype=torch.float64):
        super().__init__()
        P = torch.nn.Parameter
        self.hp = dict(hp)
        self.names = {}          # reference state_dict key -> tensor

        def reg(key, shape, buf=False, fill=None):
            t = torch.zeros(shape, dtype=dtype) if fill is None else torch.full(shape, fill, dtype=dtype)
            safe = key.replace(".", "__")
            if buf:
                self.register_buffer(safe, t)
            else:
                self.register_parameter(safe, P(t))
            self.names[key] = safe
            return safe

        # FFNN_pre.py:18-45 : Sequential [Linear, ReLU, Dropout] * n
        self.ffnn = []
        fin = in_features_FFNN
        for i in range(hp["FFNN_n_layers"]):
            fout = hp[f"FFNN_n_units_l{i}"]
            self.ffnn.append((reg(f"FFNN.model.{3*i}.weight", (fout, fin)),
                              reg(f"FFNN.model.{3*i}.bias", (fout,)),
                              float(hp.get(f"FFNN_dropout_l{i}", 0.0))))
            fin = fout
        self.d0 = fin
        # CNN_pre.py:24-60 : Sequential [Conv1d, BN, ReLU, MaxPool(10,2), Dropout] * n
        self.cnn = []
        cin, L = 4, 256
        for i in range(hp["CNN_n_layers"]):
            cout, k = hp[f"CNN_out_channels_l{i}"], hp[f"CNN_kernel_size_l{i}"]
            self.cnn.append(dict(
                w=reg(f"CNN.CNN_model.{5*i}.weight", (cout, cin, k)),
                b=reg(f"CNN.CNN_model.{5*i}.bias", (cout,)),
                g=reg(f"CNN.CNN_model.{5*i+1}.weight", (cout,), fill=1.0),
                beta=reg(f"CNN.CNN_model.{5*i+1}.bias", (cout,)),
                rm=reg(f"CNN.CNN_model.{5*i+1}.running_mean", (cout,), buf=True),
                rv=reg(f"CNN.CNN_model.{5*i+1}.running_var", (cout,), buf=True, fill=1.0),
                k=k, p=float(hp.get(f"CNN_dropout_l{i}", 0.0))))
            cin, L = cout, conv_out_len(L, k)
        self.d1 = cin * L
        c = hp["EMBRACENET_embracement_size"]
        self.c = c
        self.dock = [(reg("embracenet.docking_0.weight", (c, self.d0)), reg("embracenet.docking_0.bias", (c,))),
                     (reg("embracenet.docking_1.weight", (c, self.d1)), reg("embracenet.docking_1.bias", (c,)))]
        self.post = []
        fin = c
        for i in range(hp["n_post_layers"]):
            fout = hp[f"EMBRACENET_n_units_l{i}"]
            self.post.append((reg(f"post.{3*i}.weight", (fout, fin)), reg(f"post.{3*i}.bias", (fout,)),
                              float(hp.get(f"EMBRACENET_dropout_l{i}", 0.0)), True))
            fin = fout
        n = hp["n_post_layers"]
        self.post.append((reg(f"post.{3*n}.weight", (n_classes, fin)), reg(f"post.{3*n}.bias", (n_classes,)),
                          0.0, False))
        s = hp["selection_probabilities_FFNN"]
        self.sel = torch.tensor([s, 1.0 - s])            # fp32, EmbraceNetMultimodal.py:157
        self.last = {}

    def tensor(self, key):
        return getattr(self, self.names[key])

    def set_tensors(self, fn):
        """fn(reference_key, shape) -> numpy array; fills every parameter (not BN running stats)."""
        with torch.no_grad():
            for key, safe in self.names.items():
                t = getattr(self, safe)
                if "running_" in key:
                    continue
                t.copy_(torch.from_numpy(np.asarray(fn(key, tuple(t.shape)))).to(t.dtype))

    # ------------------------------------------------------------------ forward
    def pre_nets(self, x1, x2):
        h = x1
        for w, b, p in self.ffnn:
            h = F.relu(F.linear(h, getattr(self, w), getattr(self, b)))
            if self.training and p > 0:
                h = F.dropout(h, p, True)
        g = x2
        for L in self.cnn:
            g = F.conv1d(g, getattr(self, L["w"]), getattr(self, L["b"]), padding=(L["k"] - 1) // 2)
            g = F.batch_norm(g, getattr(self, L["rm"]), getattr(self, L["rv"]), getattr(self, L["g"]),
                             getattr(self, L["beta"]), self.training, 0.1, 1e-5)
            g = F.max_pool1d(F.relu(g), 10, 2)
            if self.training and L["p"] > 0:
                g = F.dropout(g, L["p"], True)
        return h, g.reshape(g.shape[0], -1)

    def forward(self, x, is_training=False, embracenet_dropout=True, generator=None, inject=None):
        """inject = dict(t=int64 [B] or None, u=fp64 [B, c]): the random draws of :178-182 / :84 handed in instead of taken from
        the torch generator (tests replaying the engine's device RNG, csrc/philox.h); the arithmetic is unchanged."""
        x1, x2 = x
        B = x1.shape[0]
        h0, h1 = self.pre_nets(x1, x2)
        avail = torch.ones(B, 2)
        r = t = None
        if inject is not None:
            t = inject.get("t")
            if t is not None:
                avail = F.one_hot(torch.as_tensor(t, dtype=torch.int64), 2).float()
        elif is_training and embracenet_dropout:                    # :178-182
            r = torch.rand(1, generator=generator)[0]
            if r >= 0.5:
                t = torch.round(torch.rand([B], generator=generator)).to(torch.int64)
                avail = F.one_hot(t, 2).float()
        p = self.sel.repeat(B, 1) * avail                           # :184, :73
        p = p / p.sum(-1, keepdim=True)                             # :75-76
        if not bool(torch.isfinite(p).all()):
            raise RuntimeError("invalid multinomial distribution (encountering probability entry = infinity or NaN)")
        cdf0 = p[:, 0] / (p[:, 0] + p[:, 1])                        # ATen multinomial cdf, fp32
        if inject is not None:
            u = torch.as_tensor(inject["u"], dtype=torch.float64).view(B, self.c)
        else:
            u = torch.rand(B * self.c, dtype=torch.float64, generator=generator).view(B, self.c)
        idx = cdf0.double()[:, None] < u                            # True -> modality 1
        D0 = F.relu(F.linear(h0, *[getattr(self, n) for n in self.dock[0]]))
        D1 = F.relu(F.linear(h1, *[getattr(self, n) for n in self.dock[1]]))
        E = torch.where(idx, D1, D0)                                # :80-88
        self.last = dict(r=r, t=t, idx=idx.to(torch.int64), E=E, h0=h0, h1=h1, u=u)
        y = E
        for w, b, pdrop, relu in self.post:
            y = F.linear(y, getattr(self, w), getattr(self, b))
            if relu:
                y = F.relu(y)
                if self.training and pdrop > 0:
                    y = F.dropout(y, pdrop, True)
        return y


def class_weight_tensor(target):
    """utils/utils.py:121-140 + reorder at training_models_multimodal.py:141 -> [w_neg, w_pos]."""
    t = target.reshape(-1)
    pos = int((t == 1).sum())
    neg = int((t == 0).sum())
    pi = 1.0 / pos if pos else 0.0
    ni = 1.0 / neg if neg else 0.0
    return torch.tensor([ni / (ni + pi), pi / (ni + pi)])


def batch_loss(output, target):
    """training_models_multimodal.py:141,151-154 -- fp32 weighted CE on output.float()."""
    w = class_weight_tensor(target).float()
    return F.cross_entropy(output.float(), target.reshape(-1), weight=w)


def sk_batch_ap(output, target):
    """utils/utils.py:80-86 (sklearn on hard predictions)."""
    from sklearn.metrics import average_precision_score
    pred = torch.argmax(output, dim=1).detach().numpy()
    res = average_precision_score(target.detach().numpy(), pred)
    return res if not np.isnan(res) else 0


def train_step(model, optimizer, x1, x2, target, generator=None):
    """One pass of training_models_multimodal.py:132-163 (incl. the two host syncs)."""
    optimizer.zero_grad()
    out = model([x1.double(), x2.double()], is_training=True, generator=generator)
    loss = batch_loss(out, target)
    loss.backward()
    optimizer.step()
    return loss.item(), sk_batch_ap(out, target)


def eval_step(model, x1, x2, target, generator=None):
    """One pass of training_models_multimodal.py:167-192."""
    out = model([x1.double(), x2.double()], generator=generator)
    loss = batch_loss(out, target)
    return loss.item(), sk_batch_ap(out, target), out
