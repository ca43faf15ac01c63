"""Train / evaluate / tune harness with the reference's call surface
(BIOINF_tesi/models/utils/training_models_multimodal.py): ``fit_multimodal`` (:40-226),
``Param_Search_Multimodal`` (:232-462), ``Kfold_CV_Multimodal`` (:475-798).

What changes relative to the reference is only *where* the step runs:
  * forward / loss / backward / optimizer are HIP kernels enqueued on one stream;
  * nothing is read back per step -- the reference's ``loss.item()`` (:160) and sklearn AUPRC (:162) are
    replaced by a device-resident table of (loss, TP, PP, P, n) per step, fetched once per epoch and
    turned into the same scores by closed forms (metrics.py);
  * with torch.distributed initialised (one process per GPU, RCCL) every global batch is sharded by rows
    across ranks (dist.py).
Orchestration that never touches the device (Optuna study handling, the CV split, data preparation) is
host code kept for API compatibility; its third-party dependencies are imported lazily.
"""
import copy
import os
from collections import defaultdict

import numpy as np
import torch

from . import dist as D
from . import functional as F_
from . import metrics as M
from . import optim as fused_optim

CELL_LINES = ['A549', 'GM12878', 'H1', 'HEK293', 'HEPG2', 'K562', 'MCF7']
TASKS = ['active_E_vs_inactive_E', 'active_P_vs_inactive_P', 'active_E_vs_active_P',
         'inactive_E_vs_inactive_P', 'active_EP_vs_inactive_rest']

# The reference always trains in fp64 (model.double(), :115/:328).  "float32" and "bfloat16" are the
# fast settings of this engine (bf16 = bf16 activations/weight shadows, fp32 masters and accumulation).
DEFAULT_PRECISION = "float64"
_PRECISIONS = {"float64": torch.float64, "float32": torch.float32, "bfloat16": torch.bfloat16}


def set_default_precision(name):
    global DEFAULT_PRECISION
    if name not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}")
    DEFAULT_PRECISION = name


def _check_names(cell_line, task):
    if cell_line not in CELL_LINES:
        raise ValueError(f"Argument 'cell_line' has an incorrect value: use one among {CELL_LINES}")
    if task not in TASKS:
        raise ValueError(f"Argument 'task' has an incorrect value: use one among {TASKS} ")


def prepare_model(model, device, precision=None):
    """model.double().to(device) of the reference (:115), generalised to the engine's precisions."""
    precision = precision or DEFAULT_PRECISION
    dt = _PRECISIONS[precision]
    if dt == torch.bfloat16:
        model = model.float()
        if hasattr(model, "set_compute_dtype"):
            model.set_compute_dtype(torch.bfloat16)
    else:
        model = model.to(dt)
        if hasattr(model, "set_compute_dtype"):
            model.set_compute_dtype(None)
    return model.to(device)


def _input_dtype(model):
    return next(model.parameters()).dtype


def _is_embracenet(model):
    return type(model).__name__ == 'EmbraceNetMultimodal'    # the reference switches on this string (:146)


def fused_ticks(model, optimizer, device):
    """Hands the per-step counters (the model's RNG step, the fused optimizer's update count) over to the loss kernel:
    returns the tensors to pass as ``weighted_ce_with_grad(ticks=...)`` and switches the owners' own one-thread
    counter launches off (two launches fewer per training step)."""
    ticks = []
    if hasattr(model, "step_counter"):
        model.defer_step_tick = True
        ticks.append(model.step_counter(device))
    if optimizer is not None and hasattr(optimizer, "step_counter"):
        optimizer.external_tick = True
        ticks.append(optimizer.step_counter(device))
    return tuple(ticks)


GRAPH_STEPS = False          # default of StepRunner(graph=None): replay whole steps as hipGraphs (set_graph_steps)
_GRAPH_WARMUP = 2            # eager steps of a batch shape before it is captured (allocations, optimizer state, packs)
_GRAPH_LIMIT = 24            # graphs per StepRunner (shapes x hyper-parameter values); beyond that new keys run eagerly


def set_graph_steps(enable):
    """Let fit_multimodal / Kfold_CV_Multimodal replay whole train / eval steps as hipGraphs (one per batch shape) instead
    of launching their ~26 kernels from Python every batch.  Needs the device RNG (``rng_mode == "philox"``), a fused
    optimizer of this package and a single process; anything else keeps running eagerly.  Results are identical."""
    global GRAPH_STEPS
    GRAPH_STEPS = bool(enable)


class StepRunner:
    """One train or eval step of training_models_multimodal.py:132-163 / :167-192 on device tensors."""

    def __init__(self, model, optimizer, device, graph=None):
        self.model, self.optimizer, self.device = model, optimizer, device
        self.graph = GRAPH_STEPS if graph is None else bool(graph)
        self._graphs = {}
        self.fuse_loss = True
        self.consume_slabs = True      # let this package's optimizers sum the backward's gradient slabs in their own launch
        on_gpu_ = torch.device(device).type == "cuda"
        # data parallel on the GPU: every gradient is a view of one of two flat buffers the backward kernels write directly;
        # the first bucket (post stack, head, docking) is all-reduced while the pre-networks' backward still runs
        self.flat = D.BucketedFlatGrads(model) if (optimizer is not None and D.world_size() > 1 and on_gpu_
                                                   and next(model.parameters()).is_cuda) else None
        self.bucket = D.GradBucket(model.parameters()) if (optimizer is not None and self.flat is None) else None
        self._redundant = False
        self.counts = torch.zeros(2, dtype=torch.int64, device=device)
        on_gpu = torch.device(device).type == "cuda"
        self.model_tick = model.step_counter(device) if (on_gpu and hasattr(model, "step_counter")) else None
        self.opt_tick = optimizer.step_counter(device) if (on_gpu and optimizer is not None and hasattr(optimizer, "step_counter")) else None

    def _shard(self, x_1, x_2, target):
        world = D.world_size()
        B = x_1.shape[0]
        self._redundant = False
        if world == 1:
            return x_1, x_2, target, 0
        if B < world:
            # fewer rows than ranks (a ragged last batch): an empty shard cannot run the kernels and must not leave its
            # peers waiting in a collective.  Every rank knows B, so every rank takes the same branch: all process the whole
            # (tiny) batch, and only rank 0's loss share, counts and gradients enter the sums (_drop_redundant).
            self._redundant = True
            return x_1, x_2, target, 0
        row0, rows = D.shard_rows(B, D.rank(), world)
        sl = slice(row0, row0 + rows)
        return x_1[sl], x_2[sl], target[sl], row0

    def _drop_redundant(self, table_slots=None):
        """redundant-batch mode: ranks other than 0 contribute zeros"""
        if not self._redundant or D.rank() == 0:
            return
        if table_slots is not None:
            for t in table_slots:
                t.zero_()
        if self.flat is not None:
            self.flat.zero_()
        elif self.bucket is not None:
            for p in self.bucket.params:
                if p.grad is not None:
                    p.grad.zero_()

    # ---- hipGraph replay of whole steps ---------------------------------------------------------------------------------
    def _graphable(self, training):
        m = self.model
        if not (self.graph and torch.device(self.device).type == "cuda"):
            return False
        if D.world_size() > 1 and not (self.flat is not None and torch.distributed.get_backend() == "nccl"):
            return False                                    # collectives of other backends cannot be captured
        if getattr(m, "rng_mode", None) != "philox" or self.model_tick is None:
            return False
        return (not training) or self.opt_tick is not None

    def _graph_step(self, training, x_1, x_2, target, table):
        """Returns (output, loss) from a captured step, or None when this batch shape is still warming up / not eligible."""
        if not self._graphable(training):
            return None
        model, dev = self.model, self.device
        dt = getattr(model, "compute_dtype", None) or _input_dtype(model)
        dt2 = torch.uint8 if x_2.dtype == torch.uint8 else dt
        # launch arguments frozen at capture: shapes, dtypes, the parameter storage and the optimizer's hyper-parameters
        # (an lr scheduler therefore makes a new graph per value; after _GRAPH_LIMIT graphs the runner stops capturing)
        hyper = tuple(tuple(sorted((k, v) for k, v in g_.items() if k != "params" and isinstance(v, (int, float, bool, tuple))))
                      for g_ in self.optimizer.param_groups) if (training and self.optimizer is not None) else ()
        key = (bool(training), bool(model.training), tuple(x_1.shape), tuple(x_2.shape), dt, dt2,
               next(model.parameters()).data_ptr(), hyper)
        ent = self._graphs.get(key)
        if ent is None:
            if len(self._graphs) >= _GRAPH_LIMIT:
                return None
            ent = self._graphs[key] = {"seen": 0}
        if ent.get("failed"):
            return None
        if "g" not in ent:
            ent["seen"] += 1
            if ent["seen"] <= _GRAPH_WARMUP:
                return None                                 # eager (the caller's normal path)
            def static(t, dtype):
                # data.device_loaders hands out reused staging buffers: capture THEIR addresses (no copy per step);
                # anything else (host batches, a list of device batches) is copied into a buffer owned by the graph
                if getattr(t, "_emb_staging", False) and t.dtype == dtype and t.device == torch.device(dev) and t.is_contiguous():
                    return t
                return torch.empty(tuple(t.shape), dtype=dtype, device=dev).copy_(t, non_blocking=True)
            st = ent["static"] = dict(x1=static(x_1, dt), x2=static(x_2, dt2), table=M.StepTable(1, dev))
            st["y"] = static(target, torch.int64).reshape(-1)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g):
                    st["table"].n = 0
                    if training:
                        st["out"], st["loss"] = self._train_step_eager(st["x1"], st["x2"], st["y"], st["table"], set_to_none=True)
                    else:
                        st["out"], st["loss"] = self._eval_step_eager(st["x1"], st["x2"], st["y"], st["table"])
            except Exception as e:                          # something in this configuration does not capture: stay eager
                import warnings
                warnings.warn(f"StepRunner: step capture failed ({type(e).__name__}: {e}); this batch shape runs eagerly")
                F_.reduce_defer(False)
                ent["failed"] = True
                ent.pop("static", None)
                torch.cuda.synchronize()
                return None
            # keep only the values: holding the autograd graph of the captured step alive would pin its AccumulateGrad nodes
            # to this capture's stream and make every later capture synchronise with it
            st["out"], st["loss"] = st["out"].detach(), st["loss"].detach()
            ent["g"] = g                                    # (capture does not execute: replay below runs this batch)
        st = ent["static"]
        for buf, t in ((st["x1"], x_1), (st["x2"], x_2), (st["y"], target)):
            if not (t.is_cuda and t.data_ptr() == buf.data_ptr() and t.dtype == buf.dtype):
                buf.copy_(t.reshape(buf.shape), non_blocking=True)
        ent["g"].replay()
        loss_slot, count_slot = table.slot()
        loss_slot.copy_(st["table"].loss[:1])
        count_slot.copy_(st["table"].counts[0])
        return st["out"], st["loss"]

    def _forward_loss(self, x_1, x_2, target, training, table):
        model, dev = self.model, self.device
        x_1, x_2, target, row0 = self._shard(x_1, x_2, target)
        dt = getattr(model, "compute_dtype", None) or _input_dtype(model)   # cast rides on the host->device copy
        x_1 = x_1.to(dev, dtype=dt, non_blocking=True)
        x_2 = x_2.to(dev, non_blocking=True) if x_2.dtype == torch.uint8 else x_2.to(dev, dtype=dt, non_blocking=True)   # codes stay bytes
        tgt = target.to(dev, non_blocking=True).reshape(-1)
        if hasattr(model, "rng_row0"):
            model.rng_row0 = row0
        global_counts = D.world_size() > 1 and not self._redundant
        if global_counts:                                   # class weights of the GLOBAL batch
            F_.count_labels(tgt, out=self.counts)
            D.allreduce_counts(self.counts)
        prev = getattr(model, "defer_step_tick", False)
        if self.model_tick is not None:
            model.defer_step_tick = True                    # the loss kernel below advances the RNG step
        loss_slot, count_slot = table.slot()
        ticks = (self.model_tick, self.opt_tick if training else None)
        # fp32 / bf16 runs: the classifier head takes the loss (and its own backward) into its launch (csrc/head.hip)
        fused = self.fuse_loss and hasattr(model, "fused_loss_ready") and model.fused_loss_ready(x_1.shape[0])
        if fused:
            model.arm_fused_loss(F_.FusedLoss(tgt, self.counts, global_counts, loss_slot, count_slot, ticks))
        sync_mods = [m for m in model.modules() if getattr(m, "sync_batchnorm", False)] if self._redundant else []
        for m in sync_mods:
            m.sync_batchnorm = False                        # every rank holds the whole batch: nothing to exchange
        try:
            if training and _is_embracenet(model):
                output = model([x_1, x_2], is_training=True)
            else:
                output = model([x_1, x_2])
        finally:
            for m in sync_mods:
                m.sync_batchnorm = True
            if self.model_tick is not None:
                model.defer_step_tick = prev
            if fused:
                model.arm_fused_loss(None)
        if fused:
            return output, loss_slot.view(()), None
        loss, dlogits = F_.weighted_ce_with_grad(output, tgt, class_counts=self.counts, global_counts=global_counts,
                                                 confusion=count_slot, loss_out=loss_slot, ticks=ticks)
        return output, loss, dlogits

    def train_step(self, x_1, x_2, target, table):
        done = self._graph_step(True, x_1, x_2, target, table)
        return done if done is not None else self._train_step_eager(x_1, x_2, target, table)

    def eval_step(self, x_1, x_2, target, table):
        done = self._graph_step(False, x_1, x_2, target, table)
        return done if done is not None else self._eval_step_eager(x_1, x_2, target, table)

    def _train_step_eager(self, x_1, x_2, target, table, set_to_none=None):
        if self.flat is not None:
            pass                        # gradients live in the flat buckets and are overwritten by every backward pass
        elif set_to_none is None:
            self.optimizer.zero_grad()
        else:
            self.optimizer.zero_grad(set_to_none=set_to_none)
        slots_before = table.n
        output, loss, dlogits = self._forward_loss(x_1, x_2, target, True, table)
        deferred = output.is_cuda
        if deferred:
            F_.reduce_defer(True)       # the per-layer slab reductions of the backward are queued ...
        if self.flat is not None and not self._redundant:
            def early():                # fusion layer's backward is enqueued: its slabs, then the first bucket's all-reduce
                F_.reduce_flush()
                self.flat.allreduce_early()
            F_.set_after_embrace_backward(early)
        # single process with one of this package's optimizers: its launch sums the queued slabs itself (no reduction launch);
        # otherwise (torch optimizers, data parallel: the all-reduce needs finished gradients) they are flushed after backward
        consume = (deferred and self.consume_slabs and self.flat is None and D.world_size() == 1
                   and isinstance(self.optimizer, fused_optim._Fused))
        try:
            # the loss is the root of the graph: d loss / d logits comes from the loss kernel (or, fused, is already inside
            # the head's node, which ignores what it is handed)
            output.backward(dlogits if dlogits is not None else output.detach())
        except BaseException:
            F_.set_after_embrace_backward(None)
            if deferred:
                F_.reset()              # whatever this stream has parked points at tensors of a step that will not finish
            raise
        F_.set_after_embrace_backward(None)
        if deferred:
            F_.reduce_defer(False)
            if not consume:
                F_.reduce_flush()       # ... and run in one launch here
        if self._redundant:
            self._drop_redundant((table.loss[slots_before:table.n], table.counts[slots_before:table.n]))
        if self.flat is not None:
            self.flat.finish()
        else:
            self.bucket.allreduce()
        if self.opt_tick is not None:
            self.optimizer.external_tick = True             # already advanced by the loss kernel of this step
        try:
            self.optimizer.step()
        except BaseException:
            if deferred:
                F_.reset()
            raise
        finally:
            if self.opt_tick is not None:
                self.optimizer.external_tick = False
        if consume:
            F_.reduce_flush()           # whatever the optimizer launch did not take (normally nothing: no launch)
        return output, loss

    def _eval_step_eager(self, x_1, x_2, target, table):
        slots_before = table.n
        with torch.no_grad():
            output, loss, _ = self._forward_loss(x_1, x_2, target, False, table)
        if self._redundant:
            self._drop_redundant((table.loss[slots_before:table.n], table.counts[slots_before:table.n]))
        return output, loss


def _epoch_scores(table, n_batches_for_mean):
    """losses / counts table -> (sum of losses, mean AUPRC, mean macro P/R/F1) with the reference's
    divisor len(loader['FFNN']) (:195-197)."""
    losses, counts = table.fetch()
    if D.world_size() > 1:                                   # shards -> global batches
        t = torch.from_numpy(np.concatenate([counts.reshape(-1), (losses * 1e9).astype(np.int64)]))
        dev = "cuda" if torch.cuda.is_available() and torch.distributed.get_backend() == "nccl" else "cpu"
        t = t.to(dev)
        torch.distributed.all_reduce(t)
        t = t.cpu().numpy()
        counts, losses = t[:counts.size].reshape(-1, 4), t[counts.size:] / 1e9
    ap = sum(M.ap_from_counts(*map(int, c)) for c in counts)
    prf = sum((M.prf_from_counts(*map(int, c)) for c in counts), np.zeros(3))
    return float(losses.sum()), ap / n_batches_for_mean, prf / n_batches_for_mean


def _pairs(loader):
    for load1, load2 in zip(loader['FFNN'], loader['CNN']):
        x_1, target = load1
        x_2, _t = load2
        assert (len(x_1) == len(x_2))
        if not target.is_cuda:                               # the reference's alignment assert (:136); on host only
            assert (torch.eq(target, _t).all())
        yield x_1, x_2, target


def fit_multimodal(model, train_loader, test_loader, device, cell_line, task, optimizer=None, num_epochs=100,
                   patience=4, delta=0, verbose=True, checkpoint_path=None, precision=None, graph=None):
    """Train `model`, or reload it and its scores when `checkpoint_path` exists.  `precision` / `graph` are this engine's
    additions (see prepare_model / set_graph_steps); everything else is the reference's signature.
    Returns (AUPRC_train_scores, AUPRC_test_scores, F1_precision_recall_test_scores), one entry per epoch."""
    _check_names(cell_line, task)
    if checkpoint_path is not None and os.path.exists(checkpoint_path):
        checkpoint = torch.load(checkpoint_path, map_location=torch.device(device), weights_only=False)
        model.load_state_dict(checkpoint['model_state_dict'])
        return (checkpoint['AUPRC_train_scores'], checkpoint['AUPRC_test_scores'],
                checkpoint['F1_precision_recall_test_scores'])
    if optimizer is None:
        raise ValueError("fit_multimodal needs an optimizer built on model.parameters()")

    AUPRC_train_scores, AUPRC_test_scores, F1_precision_recall_test_scores = [], [], []
    model = prepare_model(model, device, precision)
    early_stopping = M.EarlyStopping(patience=patience, delta=delta, verbose=True)
    runner = StepRunner(model, optimizer, device, graph=graph)
    cap = max(len(train_loader['FFNN']), len(test_loader['FFNN'])) + 2   # BalancePos sampler yields len+1 batches
    table = M.StepTable(cap, device)

    try:
        from tqdm.auto import tqdm
        epochs = tqdm(range(1, num_epochs + 1), desc='Epochs')
    except Exception:                                        # pragma: no cover
        epochs = range(1, num_epochs + 1)
    for epoch in epochs:
        model.train()
        for x_1, x_2, target in _pairs(train_loader):
            runner.train_step(x_1, x_2, target, table)
        train_loss, AUPRC_train, _ = _epoch_scores(table, len(train_loader['FFNN']))

        model.eval()
        for x_1, x_2, target in _pairs(test_loader):
            runner.eval_step(x_1, x_2, target, table)
        test_loss, AUPRC_test, F1_precision_recall_test = _epoch_scores(table, len(test_loader['FFNN']))

        AUPRC_train_scores.append(AUPRC_train)
        AUPRC_test_scores.append(AUPRC_test)
        F1_precision_recall_test_scores.append(F1_precision_recall_test)
        if verbose is True and D.rank() == 0:
            print('Epoch: {} \tTraining AUPRC score: {:.4f} \tTest AUPRC score: {:.4f} \tTraining Loss: {:.4f} '
                  '\tTest Loss: {:.4f}'.format(epoch, AUPRC_train, AUPRC_test, train_loss, test_loss))
        early_stopping(AUPRC_test)
        if early_stopping.early_stop:
            print('Early stopping the training')
            break

    D.broadcast_buffers(model)                               # BatchNorm running statistics: every replica ends with rank 0's
    if checkpoint_path:
        if D.rank() == 0:                                    # one writer; the others wait for the file to be complete
            torch.save({'model_state_dict': model.state_dict(), 'AUPRC_train_scores': AUPRC_train_scores,
                        'AUPRC_test_scores': AUPRC_test_scores,
                        'F1_precision_recall_test_scores': F1_precision_recall_test_scores}, checkpoint_path)
        D.barrier()
    return AUPRC_train_scores, AUPRC_test_scores, F1_precision_recall_test_scores


def make_optimizer(name, params, lr, weight_decay):
    """The three optimizers of the reference's search space (:318-325) as fused HIP kernels."""
    return getattr(fused_optim, name)(params, lr=lr, weight_decay=weight_decay)


class Param_Search_Multimodal:
    """Optuna hyper-parameter search (training_models_multimodal.py:232-462).  `model` is the model CLASS;
    each trial builds it from the trial, trains with per-epoch pruning and saves it whole."""

    def __init__(self, model, train_loader, test_loader, num_epochs, study_name, device, cell_line, task,
                 sampler='TPE', n_trials=3, storage='BIOINF_optuna_tuning.db', precision=None):
        _check_names(cell_line, task)
        self.model_ = copy.deepcopy(model)
        self.train_loader, self.test_loader = train_loader, test_loader
        self.num_epochs, self.study_name, self.device = num_epochs, study_name, device
        self.cell_line, self.task, self.n_trials, self.storage = cell_line, task, n_trials, storage
        self.precision = precision
        self.model_name = model.__name__
        self.sampler_name = sampler

    def _sampler(self):
        import optuna
        if self.sampler_name == 'BO':
            from optuna.integration import BoTorchSampler
            return BoTorchSampler()
        if self.sampler_name == 'random':
            return optuna.samplers.RandomSampler()
        return optuna.samplers.TPESampler()

    def objective(self, trial):
        in_features_FFNN = M.get_input_size(self.train_loader['FFNN'])
        self.model = self.model_(trial, cell_line=self.cell_line, task=self.task, device=self.device,
                                 in_features_FFNN=in_features_FFNN)
        optimizer_name = trial.suggest_categorical("optimizer", ["Nadam", "Adam", "RMSprop"])
        suggest_log = getattr(trial, "suggest_loguniform", None) or \
            (lambda n, lo, hi: trial.suggest_float(n, lo, hi, log=True))
        lr = suggest_log("lr", 1e-5, 1e-1)
        weight_decay = suggest_log("weight_decay", 1e-4, 1e-1)
        self.model = prepare_model(self.model, self.device, self.precision)
        optimizer = make_optimizer(optimizer_name, self.model.parameters(), lr, weight_decay)
        early_stopping = M.EarlyStopping(patience=4, verbose=True)
        runner = StepRunner(self.model, optimizer, self.device)
        table = M.StepTable(max(len(self.train_loader['FFNN']), len(self.test_loader['FFNN'])) + 2, self.device)
        AUPRC_test = 0.0
        for epoch in range(1, self.num_epochs + 1):
            self.model.train()
            for x_1, x_2, target in _pairs(self.train_loader):
                runner.train_step(x_1, x_2, target, table)
            table.fetch()
            self.model.eval()
            for x_1, x_2, target in _pairs(self.test_loader):
                runner.eval_step(x_1, x_2, target, table)
            _, AUPRC_test, _ = _epoch_scores(table, len(self.test_loader['FFNN']))
            trial.report(AUPRC_test, epoch)
            if trial.should_prune():
                import optuna
                raise optuna.exceptions.TrialPruned()
            early_stopping(AUPRC_test)
            if early_stopping.early_stop:
                print('Early stopping the training')
                break
        torch.save(self.model, f'{self.study_name}{trial.number}.pt')
        return AUPRC_test

    def run_trial(self):
        import optuna
        study = optuna.create_study(study_name=self.study_name, direction="maximize",
                                    pruner=optuna.pruners.PatientPruner(optuna.pruners.MedianPruner(), patience=2),
                                    storage=f'sqlite:///{self.storage}', load_if_exists=True, sampler=self._sampler())
        done = lambda st: [t for t in study.trials if t.state == st]
        complete = done(optuna.trial.TrialState.COMPLETE)
        if len(complete) < self.n_trials:
            self.n_trials -= len(complete)
            study.optimize(self.objective, n_trials=self.n_trials)
        self.best_model = torch.load(f'{self.study_name}{study.best_trial.number}.pt', torch.device(self.device),
                                     weights_only=False)
        print("Study statistics: ")
        print("  Number of finished trials: ", len(study.trials))
        print("  Number of pruned trials: ", len(done(optuna.trial.TrialState.PRUNED)))
        print("  Number of complete trials: ", len(done(optuna.trial.TrialState.COMPLETE)))
        trial = study.best_trial
        self.best_params = trial.params
        print("Best trial:\n  Value: ", trial.value, "\n  Params: ")
        for key, value in trial.params.items():
            print("    {}: {}".format(key, value))


def dd():
    return defaultdict(list)


def path_augmentation(augmentation):
    return '_augmentation' if augmentation else ''


def get_imbalance(y=None, n_pos=None, n_neg=None, n_decim=3):
    """positives / negatives rounded to `n_decim` decimals -- the quantity the reference compares with its rebalance
    threshold (BIOINF_tesi/data_pipe/utils.py:280-306; used at training_models_multimodal.py:533).  A ratio, not a
    fraction: 10 % positives give 0.111, a positives-majority split gives > 1 (never re-balanced).  No negatives is a
    ZeroDivisionError, as in the reference."""
    if y is not None:
        y = _rows(y).reshape(-1)
        n_pos, n_neg = int((y == 1).sum()), int((y == 0).sum())
    return float(np.round(float(n_pos / n_neg), n_decim))


def _rows(a):
    """pandas objects / lists of them / arrays -> numpy rows (row order = concatenation order)"""
    if isinstance(a, (list, tuple)):
        return np.concatenate([_rows(x) for x in a])
    return a.to_numpy() if hasattr(a, "to_numpy") else np.asarray(a)


def encode_sequences(seq):
    """Nucleotide windows -> uint8 base codes [N, L] (a, c, g, t = 0..3 -- the channel order of the reference's one-hot
    encoder, data_pipe/utils.py:269-276 -- anything else, e.g. 'n', = 4: an all-zero column).  Accepts strings, one-hot
    [N, 4, L] arrays or ready-made codes."""
    a = _rows(seq)
    if a.dtype == np.uint8 and a.ndim == 2:
        return a
    if a.ndim == 3:
        return F_.pack_onehot(torch.as_tensor(a)).cpu().numpy()
    strings = [str(x[0] if isinstance(x, (np.ndarray, list, tuple)) else x).lower() for x in a]
    lut = np.full(256, 4, dtype=np.uint8)
    for i, ch in enumerate("acgt"):
        lut[ord(ch)] = i
    width = max(len(t) for t in strings)
    out = np.full((len(strings), width), 4, dtype=np.uint8)
    for r, t in enumerate(strings):
        out[r, :len(t)] = lut[np.frombuffer(t.encode("ascii", "replace"), dtype=np.uint8)]
    return out


class Kfold_CV_Multimodal:
    """k-fold cross-validation driver with the reference's call surface (training_models_multimodal.py:475-798): per fold,
    tune on a train / validation split, re-initialise the best model and train / test it.

    What is built here is the DEVICE side: every split is staged once in HBM and batched by row gathers
    (data.device_loaders -- the balanced training sampler and the shuffled test loader reproduce the reference's batch
    index lists bit for bit, fixture G11).  The reference's table ETL around it (SMOTE / MICE re-balancing and
    augmentation, BIOINF_tesi/data_pipe) is out of scope: pass `rebalance=callable(X, y, sequence,
    threshold) -> (X, y)` to apply it; without one, a training split the reference would have re-balanced raises instead
    of silently training on different data."""

    def __init__(self, rebalance=None):
        self.scores_dict = defaultdict(dd)
        self.scores_dict['final_test_AUPRC_scores'] = []
        self.scores_dict['final_train_AUPRC_scores'] = []
        self.model_, self.optimizer = [], []
        self.best_params = defaultdict(dict)
        self.rebalance = rebalance

    def build_dataloaders_forCV(self, X_1, X_2, y, batch_size=100, training=True, augmentation=False):
        """{'FFNN': loader, 'CNN': loader} over one device-resident split (the reference builds the two loaders separately
        with one sampler seed, :592-616; here they share the staged split and its index lists)."""
        from . import data
        y_rows = _rows(y).reshape(-1)
        if training:
            needs = augmentation or get_imbalance(y_rows) < self.rebalance_threshold    # the reference's trigger (:533)
            if needs:
                if self.rebalance is None:
                    raise NotImplementedError(
                        "this training split is below the rebalance threshold (or augmentation was requested): the reference "
                        "re-balances it with SMOTE/MICE (data_pipe.utils.data_rebalancing / data_augmentation), which is outside "
                        "this engine -- construct Kfold_CV_Multimodal(rebalance=...) with that step")
                X_1, y1 = self.rebalance(X_1, y, False, self.rebalance_threshold)
                X_2, y = self.rebalance(X_2, y, True, self.rebalance_threshold)
                y_rows = _rows(y).reshape(-1)
        x1 = _rows(X_1).astype(np.float64)
        seq = encode_sequences(X_2)
        dt = _PRECISIONS[self.precision or DEFAULT_PRECISION]
        if training:
            return data.device_loaders(x1, seq, y_rows, batch_size, self.device, balanced=True, random_state=self.random_state,
                                       feature_dtype=dt)
        return data.device_loaders(x1, seq, y_rows, batch_size * 2, self.device, balanced=False,
                                   random_state=self.random_state + 30, feature_dtype=dt)

    def hyper_tuning(self, train_loader, test_loader, num_epochs, cell_line, task, study_name, device, sampler):
        param_search = Param_Search_Multimodal(model=self.model_, train_loader=train_loader, test_loader=test_loader,
                                               num_epochs=num_epochs, cell_line=cell_line, task=task, device=device,
                                               sampler=sampler, n_trials=3, study_name=study_name,
                                               precision=self.precision)
        param_search.run_trial()
        best_params = param_search.best_params
        self.model_ = param_search.best_model
        self.best_params[self.i] = best_params
        self.model_.apply(M.weight_reset)
        self.optimizer = make_optimizer(best_params['optimizer'], self.model_.parameters(), best_params['lr'],
                                        best_params['weight_decay'])

    def model_testing(self, train_loader, test_loader, num_epochs, test_model_path, device, cell_line, task,
                      checkpoint_path=None):
        AUPRC_train, AUPRC_test, other_scores = fit_multimodal(
            model=self.model_, train_loader=train_loader, test_loader=test_loader, device=device, cell_line=cell_line,
            task=task, optimizer=self.optimizer, num_epochs=num_epochs, patience=4, verbose=False,
            checkpoint_path=f'{checkpoint_path}.pt', precision=self.precision)
        it = self.scores_dict[f'iteration_n_{self.i}']
        it['AUPRC_train'], it['AUPRC_test'], it['F1_precision_recall'] = AUPRC_train, AUPRC_test, other_scores
        final_test, final_train = AUPRC_test[-1], AUPRC_train[-1]
        self.scores_dict['final_test_AUPRC_scores'].append(final_test)
        self.scores_dict['final_train_AUPRC_scores'].append(final_train)
        print(f'AUPRC test score: {final_test}\n\n')
        self.avg_score.append(final_test)
        if final_test == max(self.avg_score):
            os.makedirs('models_', exist_ok=True)
            torch.save({'model_state_dict': self.model_.state_dict(), 'model_params': self.best_params[self.i]},
                       f'models_/{test_model_path}.pt')

    def __call__(self, build_dataloader_pipeline, cell_line, device, task=None, model=None, augmentation=False,
                 rebalance_threshold=0.1, random_state=789, n_folds=3, num_epochs=100, batch_size=100,
                 study_name=None, sampler='TPE', test_model_path=None, precision=None):
        from sklearn.model_selection import train_test_split
        _check_names(cell_line, task)
        self.n_folds, self.augmentation, self.rebalance_threshold = n_folds, augmentation, rebalance_threshold
        self.random_state, self.device, self.precision = random_state, device, precision
        self.avg_score, self.hp_score = [], []
        data_class = build_dataloader_pipeline.data_class
        kf, X_1, y = data_class.return_index_data_for_cv(cell_line=cell_line, sequence=False, n_folds=n_folds,
                                                         random_state=self.random_state)
        _, X_2, _ = data_class.return_index_data_for_cv(cell_line=cell_line, sequence=True, n_folds=n_folds,
                                                        random_state=self.random_state)
        take = lambda a, idx: a.iloc[idx] if hasattr(a, "iloc") else a[idx]
        for i, (train_index, test_index) in enumerate(kf.split(X_1)):
            self.i = i + 1
            STUDY_NAME = f'{study_name}_{str(self.i)}'
            print(f'>>> ITERATION N. {self.i}')
            X_train_1, X_test_1 = take(X_1, train_index), take(X_1, test_index)
            X_train_2, X_test_2 = take(X_2, train_index), take(X_2, test_index)
            y_train, y_test = take(y, train_index), take(y, test_index)
            # (:731-741) the same split of the training part for both modalities: one call with both keeps them aligned
            X_train_1, X_val_1, X_train_2, X_val_2, y_train, y_val = train_test_split(
                X_train_1, X_train_2, y_train, test_size=1 / self.n_folds, random_state=self.random_state, shuffle=True)
            self.model_ = model
            print('\n===============> HYPERPARAMETERS TUNING')
            self.hyper_tuning(self.build_dataloaders_forCV(X_train_1, X_train_2, y_train, batch_size, True, self.augmentation),
                              self.build_dataloaders_forCV(X_val_1, X_val_2, y_val, batch_size, False, False),
                              num_epochs, cell_line, task, STUDY_NAME, device, sampler)
            print('\n===============> MODEL TESTING')
            train_loader = self.build_dataloaders_forCV([X_train_1, X_val_1], [X_train_2, X_val_2], [y_train, y_val], batch_size,
                                                        True, self.augmentation)
            test_loader = self.build_dataloaders_forCV(X_test_1, X_test_2, y_test, batch_size, False, False)
            self.model_testing(train_loader, test_loader, num_epochs, test_model_path, device, cell_line, task,
                               checkpoint_path=f'{cell_line}_{model.__name__}{path_augmentation(self.augmentation)}_{task}_{self.i}_test_')
        avg_CV_AUPRC = np.round(sum(self.avg_score) / n_folds, 5)
        self.scores_dict['average_CV_AUPRC'] = avg_CV_AUPRC
        print(f'\n{n_folds}-FOLD CROSS-VALIDATION AUPRC TEST SCORE: {avg_CV_AUPRC}')
