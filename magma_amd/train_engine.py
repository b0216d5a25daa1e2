"""Training engine: explicit forward/backward of the MAGMA graph on the HIP
kernels + fused clip/AdamW + RCCL gradient all-reduce.  Replaces the DeepSpeed
engine the reference trains with (reference train.py:103-111,
train_loop.py:7-21, config.py:124-134) behind the same object protocol:

    engine(images, captions) -> .loss ; engine.backward(loss) ; engine.step()
    engine.train() / .eval() ; .save_checkpoint / .load_checkpoint ; .lr_scheduler.get_lr()

No torch.autograd: every gradient is produced by a C-ABI kernel (dgrad = the
same MFMA GEMM on a transposed weight copy; wgrad = the same GEMM on transposed
activations; ReLU/GELU gradients fused into GEMM epilogues; flash-attention
backward; LayerNorm / CE / rotary / pool backward kernels).  Trainable set
(SURVEY Q2): adapters, CLIP trunk (conv + BN affine, BN statistics frozen, Q5),
ImagePrefix proj + LayerNorm.  288 GB of HBM: activations are kept, nothing is
recomputed -- "no recompute" policy for every reported number.
"""
from __future__ import annotations

import contextlib
import math
import os
from pathlib import Path
from typing import Dict, List, Optional

import torch
import torch.distributed as dist

from . import ops
from .language_model import LMOutput
from .ops import BF16, PackedLinear, RawWeight

F32 = torch.float32


# ----------------------------------------------------------------------------
# flat fp32 optimizer state, one block per (lr, weight_decay) group
# ----------------------------------------------------------------------------
class FlatGroup:
    def __init__(self, params: List[torch.nn.Parameter], lr: float, weight_decay: float, device):
        self.params = params
        self.lr_max, self.wd = lr, weight_decay
        self.offsets, n = [], 0
        for p in params:
            self.offsets.append(n)
            n += (p.numel() + 7) // 8 * 8          # keep every view 16-byte aligned in the bf16 copy as well
        self.n = n
        self.master = torch.zeros(n, dtype=F32, device=device)
        self.m = torch.zeros(n, dtype=F32, device=device)
        self.v = torch.zeros(n, dtype=F32, device=device)
        self.grad = torch.zeros(n, dtype=F32, device=device)
        self.model = torch.zeros(n, dtype=BF16, device=device)     # bf16 copy the kernels read
        self.comm = None                                           # bf16 exchange buckets (data-parallel runs only)
        for p, o in zip(params, self.offsets):
            self.master[o:o + p.numel()].copy_(p.data.reshape(-1).float())
        self.model.copy_(self.master)
        # re-point the module parameters at the flat bf16 buffer (zero-copy operands)
        for p, o in zip(params, self.offsets):
            p.data = self.model[o:o + p.numel()].view(p.shape)
        self.index = {id(p): k for k, p in enumerate(params)}      # parameter -> slot, O(1) (was a list scan per lookup)

    def view(self, buf: torch.Tensor, p: torch.nn.Parameter) -> torch.Tensor:
        o = self.offsets[self.index[id(p)]]
        return buf[o:o + p.numel()].view(p.shape)

    def state_by_name(self, names: Dict[int, str]) -> dict:
        """Optimizer state as UNPADDED per-parameter slices keyed by the parameter's state-dict name: the checkpoint does
        not depend on the flat buffers' padding / ordering (which changed between rounds: 4 -> 8 element alignment)."""
        out = {}
        for p, o in zip(self.params, self.offsets):
            n = p.numel()
            out[names[id(p)]] = {"master": self.master[o:o + n].cpu().clone(), "m": self.m[o:o + n].cpu().clone(),
                                 "v": self.v[o:o + n].cpu().clone()}
        return out

    def load_state_by_name(self, state: dict, names: Dict[int, str]) -> List[str]:
        missing = []
        for p, o in zip(self.params, self.offsets):
            s = state.get(names[id(p)])
            if s is None or s["master"].numel() != p.numel():
                missing.append(names[id(p)])
                continue
            n = p.numel()
            self.master[o:o + n].copy_(s["master"].reshape(-1)); self.m[o:o + n].copy_(s["m"].reshape(-1))
            self.v[o:o + n].copy_(s["v"].reshape(-1))
        return missing


class LRScheduler:
    """WarmupDecayLR / WarmupLR of DeepSpeed [UNVENDORED]: log-shaped warm-up from
    min_lr to max_lr over warmup_num_steps, then linear decay to 0 at total_num_steps."""

    def __init__(self, cfg_params: dict, kind: str, max_lrs: List[float]):
        self.kind = kind
        self.warmup = max(2, int(cfg_params.get("warmup_num_steps", 100)))
        self.total = cfg_params.get("total_num_steps")
        mins = cfg_params.get("warmup_min_lr", 0.0)
        self.min_lrs = list(mins) if isinstance(mins, (list, tuple)) else [mins] * len(max_lrs)
        self.max_lrs = list(max_lrs)
        self.last_step = 0
        self.inv_log = 1.0 / math.log(self.warmup)

    def _gamma(self, step: int) -> float:
        if step < self.warmup:
            return self.inv_log * math.log(step + 1)
        if self.kind == "WarmupDecayLR" and self.total:
            return max(0.0, (self.total - step) / max(1.0, self.total - self.warmup))
        return 1.0

    def get_lr(self) -> List[float]:
        g = self._gamma(self.last_step)
        if self.last_step < self.warmup:
            return [mn + (mx - mn) * g for mn, mx in zip(self.min_lrs, self.max_lrs)]
        return [mx * g for mx in self.max_lrs]

    def step(self):
        self.last_step += 1

    def state_dict(self):
        return {"last_step": self.last_step}

    def load_state_dict(self, sd):
        self.last_step = sd["last_step"]


_CONV_PLAN = os.environ.get("MAGMA_CONV_PLAN", "1") != "0"               # A/B switch: 0 = one re-layout / BN-fold launch per convolution
_BOTTOM_PREFIX_ONLY = os.environ.get("MAGMA_BOTTOM_PREFIX_ONLY", "1") != "0"   # A/B switch: 0 = the bottom block's input gradient on all B*S rows
_BN_GRAD_FUSED = os.environ.get("MAGMA_BN_GRAD_FUSED", "1") != "0"       # A/B switch: 0 = bn_param_grad + transpose as two passes over g
_WGRAD_INPLACE = os.environ.get("MAGMA_WGRAD_INPLACE", "1") != "0"     # A/B switch: 0 = fp32 temporary + scale_rows_acc pass


def _t(x: torch.Tensor) -> RawWeight:
    """Transposed copy of a [R, C] activation/weight as a GEMM B operand [C, R]."""
    return RawWeight(ops.transpose(x))       # [C, round_up(R,8)], zero padded


class MagmaEngine:
    """DeepSpeed-engine shim over a ``Magma`` model."""

    def __init__(self, model, config=None, param_groups: Optional[List[dict]] = None, betas=(0.9, 0.95),
                 eps: float = 1e-8, truncate: bool = False):
        self.module = model
        self.config = config or model.config
        self.device = model.device
        torch.cuda.set_device(self.device)      # kernels launch on the current device (ops._need_gpu)
        # freeze_lm: false -- every GPT-J tensor trains as well (what the published reference in fact does: magma.py:93-96
        # never sets requires_grad = False on the LM; the configs of this repo keep the paper's trainable set).
        # fp32 master + Adam moments + gradient = 16 B per parameter: ~100 GB on top of the activations at full size.
        self.lm_trainable = not self.config.freeze_lm
        self.betas, self.eps = betas, eps
        self.truncate = truncate or os.environ.get("MAGMA_TRUNCATE", "0") == "1"
        # BASELINE config[4]: the frozen-weight block GEMMs (qkv, out_proj, fc_in, fc_out; forward and dgrad) on the fp8
        # MFMA -- activations / gradients quantised per row to e4m3, weights per output channel.  Off by default.
        self.fp8 = os.environ.get("MAGMA_TRAIN_FP8", "0") == "1"
        # with self.fp8: the attention FORWARD on the fp8 MFMA as well (BASELINE config[4]: "fp8 MFMA path for GPT-J attention";
        # mg_rotary_split_fp8 + mg_attn_prefill_fp8; the backward stays bf16).  MAGMA_FP8_ATTN=0 keeps the bf16 attention.
        self.fp8_attn = os.environ.get("MAGMA_FP8_ATTN", "1") == "1"
        # with self.fp8: the two widest activations of a block never exist in bf16 -- gelu(fc_in) and its gradient leave the
        # epilogues of the fc_in GEMM / the fc_out dgrad as OCP MX e4m3 (mg_epilogue.C8: one E8M0 scale per 32 elements, local to
        # the tile that produces them) and feed fc_out / the fc_in dgrad through mg_gemm_mx_fp8 (MX-quantised weights there).
        # Removes two of the seven quantisation passes per block -- the two over [B*S, 16384] -- and the bf16 round trip of
        # both tensors.  MAGMA_TRAIN_FP8_MX=0: every fp8 GEMM on per-row scales with a quantisation pass in front (round 4).
        self.fp8_mx = os.environ.get("MAGMA_TRAIN_FP8_MX", "1") == "1"
        # with self.fp8 and self.fp8_mx: the MLP adapter's four activation-side GEMMs (down, up, and their two dgrads) on the fp8 MFMA
        # too (BASELINE config[4]: "fp8 MFMA path for GPT-J attention + adapter GEMMs"), fed through the MX output copies of the
        # producing epilogues (no quantisation pass of their own): fc_out's epilogue emits the MX copy of m -> down as an MX GEMM
        # whose ReLU epilogue emits MX t -> up consumes it; backward: the up-dgrad reads the row-quantised incoming gradient the
        # out_proj dgrad quantises anyway, its ReLU-gate epilogue emits MX dt -> the down-dgrad (+ g) emits dm as MX only, which
        # feeds the fc_out dgrad.  The adapter weights are re-quantised when they change (adapters.bump_weights_epoch).
        # Weight gradients stay bf16.  Plain ReLU MLP adapters (the MAGMA_v1 shape); MAGMA_FP8_ADAPTERS=0: bf16 adapters.
        self.fp8_adapters = os.environ.get("MAGMA_FP8_ADAPTERS", "1") == "1"
        self._ad8_cache = {}
        if self.fp8 and self.lm_trainable:
            raise NotImplementedError("MAGMA_TRAIN_FP8 quantises the FROZEN block weights once; with freeze_lm: false they change every step")
        self._fp8_packs = {}
        # MAGMA_v1 blocks: out_proj and the MLP adapter's up-projection as ONE GEMM over [ctx | t] against [W_out | W_up] (K = d + r):
        # the attention output `a` is never written or read back (2 x M x d x 2 bytes per block) and the K = 1024 up-projection no
        # longer pays a tile prologue + residual epilogue of its own.  bf16 step with a frozen LM only.  MAGMA_TRAIN_CAT=0: off.
        self.cat_up = os.environ.get("MAGMA_TRAIN_CAT", "1") == "1"
        self._out_up = {}
        self._nf_scales = {}
        self._bn_stats = {}
        # the reference's CLIP tower runs BatchNorm on its frozen statistics until the first eval phase and on
        # BATCH statistics afterwards (train.py:164,182 flips it to train mode).  Default here = frozen (steps 0 ..
        # eval_every-1 and inference); MAGMA_BN_BATCH_STATS=1 or train(bn_batch_stats=True) selects the later behaviour.
        self.bn_batch_stats = os.environ.get("MAGMA_BN_BATCH_STATS", "0") == "1"
        self._bn_dirty = False
        self._plan = self._plan_live = None     # ops.ConvOperandPlan of the CLIP trunk (frozen-statistics mode), built on first use
        self.bottom_prefix_rows = 0             # P when the last backward formed the bottom block's input gradient for the prefix rows only
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.gas = max(1, int(self.config.gradient_accumulation_steps))
        self.clip = float(self.config.gradient_clipping or 0.0)
        if param_groups is None:
            from .utils import configure_param_groups
            param_groups = configure_param_groups(model, self.config)
        self.groups: List[FlatGroup] = []
        for g in param_groups:
            self.groups.append(FlatGroup(list(g["params"]), g.get("lr", self.config.lr),
                                         g.get("weight_decay", self.config.weight_decay), self.device))
        self._where: Dict[int, tuple] = {}
        for gi, g in enumerate(self.groups):
            for p in g.params:
                self._where[id(p)] = (gi, p)
        sched = self.config.deepspeed_config_params["scheduler"]
        self.lr_scheduler = LRScheduler(sched["params"], sched["type"], [g.lr_max for g in self.groups])
        self.micro_steps = 0
        self.global_steps = 0
        self.training = True
        self._tape = None
        self._norm_sq = torch.zeros(1, dtype=F32, device=self.device)
        self._dist = dist.is_initialized()            # a 1-rank process group still exercises the overlap path
        self._comm_stream = torch.cuda.Stream(device=self.device) if self._dist else None
        # MAGMA_DP_RESERVE_CUS=k: compute on a stream that leaves k CUs to the exchange's RCCL kernels (default 0 = off)
        self._compute_stream = None
        k_res = int(os.environ.get("MAGMA_DP_RESERVE_CUS", "0"))
        if self._dist and k_res > 0:
            from .comm import ReservedCUStream
            self._compute_stream = ReservedCUStream(self.device, k_res)
        if self._dist:
            from .comm import make_exchange
            self._exchange = make_exchange(self.device)       # torch.distributed (default) or the C-ABI mg_comm_* (RCCL)
        # gradient exchange dtype: bf16 buckets (0.77 GB per step for MAGMA_v1; the reference's ZeRO-2 reduces its fp16
        # gradients the same way) or the fp32 flat buffers themselves (MAGMA_DP_GRAD_DTYPE=fp32, 1.54 GB)
        self.exchange_bf16 = self._dist and os.environ.get("MAGMA_DP_GRAD_DTYPE", "bf16") != "fp32"
        if self.exchange_bf16:
            for g in self.groups:
                g.comm = torch.zeros(g.n, dtype=BF16, device=self.device)
        self._reduced = [[] for _ in self.groups]     # per group: (lo, hi) ranges already handed to RCCL this step
        self._works = []
        self.time_comm, self._comm_events = False, []
        self._busy_events, self._busy_steps = [], 0
        self.recompute = os.environ.get("MAGMA_TRAIN_RECOMPUTE", "0") == "1" or bool(getattr(self.module.lm.config, "gradient_checkpointing", False))
        self.overlapped_elems = 0                     # gradient elements handed over DURING backward in the last step
        if self._dist and self.world > 1:
            # every replica starts from rank 0's model: trainable masters, FROZEN parameters (a random-init GPT-J differs
            # per process unless seeded) and BatchNorm statistics -- what deepspeed.initialize does for the reference
            # (train.py:103-111 broadcasts every module parameter).  One-off 12.9 GB over xGMI.
            for g in self.groups:
                self._exchange.broadcast(g.master, src=0)
                g.model.copy_(g.master)
            for p in model.parameters():
                if id(p) not in self._where and p.device.type == "cuda":
                    self._exchange.broadcast(p.data, src=0)
            for buf in model.buffers():
                if buf.is_floating_point() and buf.device.type == "cuda":
                    self._exchange.broadcast(buf, src=0)
        model.invalidate_packed()
        self._lm_train_packs = None
        self._adapters_dirty = False

    # ---- gradient exchange overlapped with backward --------------------------------
    def _is_boundary(self) -> bool:
        """True while the micro-step being back-propagated completes an accumulation window."""
        return (self.micro_steps + 1) % self.gas == 0

    def _reduce_params_async(self, params):
        """All-reduce (SUM) the flat-gradient range covering ``params`` on the comm stream as soon as
        their gradients are final: buckets leave in reverse-backward order (block 27 ... 0, then the
        prefix, then the trunk) while the earlier blocks are still being back-propagated."""
        if not self._dist or not self._is_boundary():
            return
        spans, sizes = {}, {}
        for p in params:
            if id(p) not in self._where:
                continue
            gi, _ = self._where[id(p)]
            g = self.groups[gi]
            i = g.index[id(p)]
            lo = g.offsets[i]
            hi = g.offsets[i + 1] if i + 1 < len(g.offsets) else g.n
            a, b = spans.get(gi, (lo, hi))
            spans[gi] = (min(a, lo), max(b, hi))
            sizes[gi] = sizes.get(gi, 0) + (hi - lo)
        spans = {gi: s for gi, s in spans.items() if s[1] - s[0] == sizes[gi]}   # contiguous buckets only
        todo = [(gi, lo, hi) for gi, (lo, hi) in spans.items()
                if not any(not (hi <= a or lo >= b) for a, b in self._reduced[gi])]   # overlapping an in-flight range: step()
        bufs = [self._exchange_view(gi, lo, hi) for gi, lo, hi in todo]           # bf16 casts run on the compute stream
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        for (gi, lo, hi), buf in zip(todo, bufs):
            with torch.cuda.stream(self._comm_stream):
                self._comm_stream.wait_event(ev)
                if self.time_comm:       # busy time of the exchange stream itself (bench.py: train.data_parallel)
                    e0 = torch.cuda.Event(enable_timing=True)
                    e0.record(self._comm_stream)
                w = self._exchange.all_reduce(buf)
                if w is not None:
                    self._works.append(w)
                if self.time_comm:
                    if w is not None:
                        w.wait()         # stream-side: the event below lands behind the collective
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record(self._comm_stream)
                    self._busy_events.append((e0, e1))
            self._reduced[gi].append((lo, hi))
            self.overlapped_elems += hi - lo

    def exposed_comm_ms(self, reset: bool = True) -> Optional[float]:
        """Mean time per optimizer step the compute stream spent in / waiting for the gradient exchange after backward (the
        part of the all-reduce that was NOT hidden), over the steps taken since the last call; needs time_comm = True and
        a device synchronisation by the caller.  None without a process group."""
        if not self._comm_events:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self._comm_events) / len(self._comm_events)
        if reset:
            self._comm_events = []
        return ms

    def comm_busy_ms(self, reset: bool = True) -> Optional[float]:
        """Mean time per optimizer step the exchange stream was busy with the buckets handed over DURING backward (sum of the
        per-bucket spans on that stream), over the steps since the last call; needs time_comm = True and a device
        synchronisation by the caller.  Together with exposed_comm_ms it tells a starved exchange (busy long, nothing
        exposed: fine; busy long AND exposed: the buckets wait for CUs) from a slow link."""
        if not self._busy_events:
            return None
        ms = sum(a.elapsed_time(b) for a, b in self._busy_events) / max(1, self._busy_steps)
        if reset:
            self._busy_events, self._busy_steps = [], 0
        return ms

    def _exchange_view(self, gi, lo, hi):
        """The tensor RCCL sums for flat-gradient range [lo, hi) of group gi: the bf16 bucket (filled here from the fp32
        accumulator) or the fp32 range itself."""
        g = self.groups[gi]
        if not self.exchange_bf16:
            return g.grad[lo:hi]
        return ops.cast_f32_bf16(g.grad[lo:hi], g.comm[lo:hi])

    def _finish_reduce(self):
        """Reduce whatever backward did not hand over, then join the comm stream."""
        if not self._dist:
            return
        from .comm import allreduce_grads
        self.last_overlapped_elems, self.overlapped_elems = self.overlapped_elems, 0
        for gi, g in enumerate(self.groups):
            done = sorted(self._reduced[gi])
            pos, rest = 0, []
            for a, b in done:
                if a > pos:
                    rest.append(self._exchange_view(gi, pos, a))
                pos = max(pos, b)
            if pos < g.n:
                rest.append(self._exchange_view(gi, pos, g.n))
            allreduce_grads(rest, exchange=self._exchange, always=True)
            self._reduced[gi] = []
        for w in self._works:
            w.wait()                      # stream-side join (no host block for the RCCL backend)
        self._works = []
        torch.cuda.current_stream().wait_stream(self._comm_stream)

    # ---- helpers -------------------------------------------------------------
    def close(self):
        """Release what the engine owns outside PyTorch's allocator: the RCCL communicator of the mg_comm_* exchange."""
        ex = getattr(self, "_exchange", None)
        if ex is not None:
            ex.close()
        cs = getattr(self, "_compute_stream", None)
        if cs is not None:
            cs.close()
            self._compute_stream = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def grad_of(self, p: torch.nn.Parameter) -> torch.Tensor:
        gi, _ = self._where[id(p)]
        return self.groups[gi].view(self.groups[gi].grad, p)

    def master_of(self, p: torch.nn.Parameter) -> torch.Tensor:
        gi, _ = self._where[id(p)]
        return self.groups[gi].view(self.groups[gi].master, p)

    def is_trainable(self, p) -> bool:
        return id(p) in self._where

    def _flush_bn_stats(self):
        """fp32 running statistics updated by the batch-statistics mode -> the module's buffers (checkpoints, inference)."""
        if not self._bn_dirty:
            return
        for m in self.module.image_prefix.enc.modules():
            st = self._bn_stats.get(id(m))
            if st is not None:
                m.running_mean.copy_(st[0]); m.running_var.copy_(st[1])
        self._bn_dirty = False
        self.module.image_prefix.invalidate_packed()

    def train(self, mode: bool = True, bn_batch_stats: Optional[bool] = None):
        if bn_batch_stats is not None:
            self.bn_batch_stats = bool(bn_batch_stats)
        if not mode:
            self._flush_bn_stats()
        self.training = mode
        self.module.train(mode)
        self.module.image_prefix.enc.eval()     # the module flag stays "eval": the inference path reads the running statistics
        if not mode and self._adapters_dirty:   # inference path reads packed copies of the adapters
            self.module.lm.engine.repack_adapters(self.module.lm)
            self._adapters_dirty = False
        return self

    def eval(self):
        return self.train(False)

    # ---- forward ---------------------------------------------------------------
    @contextlib.contextmanager
    def _on_compute_stream(self):
        """MAGMA_DP_RESERVE_CUS=k (multi-rank runs): the step's kernels go to a stream that leaves k CUs to the exchange
        (comm.ReservedCUStream); the caller's stream waits on both sides, so nothing outside sees the difference."""
        cs = self._compute_stream
        if cs is None:
            yield
            return
        cur = torch.cuda.current_stream(self.device)
        cs.stream.wait_stream(cur)
        with torch.cuda.stream(cs.stream):
            yield
        cur.wait_stream(cs.stream)

    def __call__(self, images, captions, dropout_mask=None, captions_host=None) -> LMOutput:
        with self._on_compute_stream():
            return self._forward_impl(images, captions, dropout_mask, captions_host)

    def backward(self, loss=None):
        with self._on_compute_stream():
            return self._backward_impl(loss)

    def step(self):
        with self._on_compute_stream():
            return self._step_impl()

    def _forward_impl(self, images, captions, dropout_mask=None, captions_host=None) -> LMOutput:
        torch.cuda.set_device(self.device)
        if not self.training:
            with torch.no_grad():
                return self.module(images, captions)
        return self.forward_train(images, captions, dropout_mask, captions_host)

    @staticmethod
    def target_index(captions_host: torch.Tensor, P: int, eos: int, S: int):
        """Index plumbing of the loss head, computed on the HOST from the captions the data loader produced there (no
        device sync in the step): flat rows b*S + j of the hidden states whose NEXT position carries a
        label, and those labels.  Mirrors build_labels (reference utils.py:334-364): position P + k holds captions[b, k]
        for k <= first EOS of the row (the EOS itself is kept), everything else is -100; shifted by one for the loss."""
        cap = captions_host[:, : S - P].to(torch.int64)
        B, T = cap.shape
        is_eos = cap == eos
        first = torch.where(is_eos.any(1), is_eos.int().argmax(1), torch.full((B,), T - 1, dtype=torch.int64))
        k = torch.arange(T)[None, :]
        valid = k <= first[:, None]                          # label positions P + k
        if P == 0:
            valid[:, 0] = False                              # position 0 has no predecessor
        b_idx, k_idx = valid.nonzero(as_tuple=True)
        rows = b_idx * S + (P + k_idx - 1)
        return rows, cap[b_idx, k_idx], first


    def _lm_packs(self):
        """Transposed copies of the frozen LM weights for the dgrad GEMMs (one-off, 12 GB)."""
        if self._lm_train_packs is None:
            eng = self.module.lm.engine
            packs = []
            for ly, blk in zip(eng.layers, self.module.lm.transformer.h):
                a, mlp = ly._src          # the wrapped GPT-J attention / MLP modules, whatever adapter type wraps them
                pk = {
                    "qkv_t": PackedLinear(torch.cat([a.q_proj.weight, a.k_proj.weight, a.v_proj.weight], 0).t().contiguous()),
                    "out_t": PackedLinear(a.out_proj.weight.t().contiguous()),
                    "fc_in_t": PackedLinear(mlp.c_fc.weight.t().contiguous()),
                    "fc_out_t": PackedLinear(mlp.c_proj.weight.t().contiguous()),
                }
                packs.append(pk)
            V, d = eng.V, eng.d
            wt = torch.zeros(d, ops.ceil_to(V, 8), dtype=BF16, device=self.device)
            wt[:, :V] = self.module.lm.lm_head.weight.t()
            self._lm_train_packs = (packs, PackedLinear(wt))
        return self._lm_train_packs

    def _cat_out_up(self, li, ly, blk):
        """[W_out | W_up] of block li as a row-major GEMM operand [d, d + r] (bias b_up), or None where the block does not have
        the shape (plain ReLU MLP adapter without LayerNorm, no attention adapter, out_proj without a bias, bf16, frozen LM).
        The W_out columns are written once; the W_up columns are refreshed from the live parameter on every forward."""
        if not self.cat_up or self.fp8 or self.lm_trainable:
            return None
        if ly.mlp_adapter is None or ly.mlp_par is not None or ly.attn_adapter is not None:
            return None
        mod = blk.mlp[1]
        a_mod, _ = ly._src
        if not getattr(mod, "plain", False) or a_mod.out_proj.bias is not None:
            return None
        up = mod.up
        d, r = up.weight.shape
        if r % 8 or d % 8:            # 16-byte aligned [:, d:] view
            return None
        buf = self._out_up.get(li)
        if buf is None or buf.shape != (d, d + r):
            buf = self._out_up[li] = torch.empty(d, d + r, dtype=BF16, device=self.device)
            buf[:, :d].copy_(a_mod.out_proj.weight.detach())
        buf[:, d:].copy_(up.weight.data)
        return RawWeight(buf, bias=self.master_of(up.bias))

    def _adapter_ops(self, mod):
        """(down, up) RawWeights on the live bf16 parameters of an Adapter-like module, biases = fp32 master views."""
        dn, up = mod.down, mod.up
        return (RawWeight(dn.weight.data, bias=self.master_of(dn.bias)),
                RawWeight(up.weight.data, bias=self.master_of(up.bias)))

    def _par_adapter_ops(self, wrapper):
        """(down, up, scale vector or None) of a ParallelAdapter / ParallelAdapterWrapper on the live parameters.  The GEMM
        epilogue computes acc * scale[n] + bias[n]; the reference (acc + b_up) * adapter_scale, so the up bias handed to the
        epilogue is pre-multiplied by the (device-resident, trainable) scale."""
        dn, up = wrapper.down, wrapper.up
        dnw = RawWeight(dn.weight.data, bias=self.master_of(dn.bias))
        sp = wrapper.adapter_scale
        if not torch.is_tensor(sp):
            assert float(sp) == 1.0
            return dnw, RawWeight(up.weight.data, bias=self.master_of(up.bias)), None
        sval = self.master_of(sp) if self.is_trainable(sp) else sp.detach().float()
        sc = sval.reshape(1).to(F32).expand(up.weight.shape[0]).contiguous()
        return dnw, RawWeight(up.weight.data, bias=(self.master_of(up.bias) * sc).contiguous()), sc

    def _ad8_ok(self, ly, mod) -> bool:
        """fp8 adapter chain: plain ReLU MLP adapter (no LayerNorm, not parallel), shapes the 256x256 fp8 kernel's MX output covers."""
        if not (self.fp8 and self.fp8_mx and self.fp8_adapters) or ly.mlp_adapter is None or ly.mlp_par is not None:
            return False
        if not getattr(mod, "plain", False):
            return False
        r, d = mod.down.weight.shape
        return r % 256 == 0 and d % 256 == 0

    def _ad8(self, li, mod):
        """e4m3 operands of the adapter's four activation-side GEMMs, rebuilt when the weights epoch moved (every optimizer step)."""
        from . import adapters
        ep = adapters._WEIGHTS_EPOCH
        c = self._ad8_cache.get(li)
        if c is not None and c[0] == ep:
            return c[1]
        wd, wu = mod.down.weight.data, mod.up.weight.data                   # [r, d], [d, r]
        packs = {"dn": ops.PackedLinearMX.of_live(wd, self.master_of(mod.down.bias)),
                 "up": ops.PackedLinearMX.of_live(wu, self.master_of(mod.up.bias)),
                 "up_t": ops.PackedLinearFP8.of_live(ops.transpose(wu)),        # dt = g W_up        (row-quantised g)
                 "dn_t": ops.PackedLinearMX.of_live(ops.transpose(wd))}         # dm = g + dt W_dn   (MX dt)
        self._ad8_cache[li] = (ep, packs)
        return packs

    def _adapter_down(self, mod, x_in, dn, sv, key):
        """t = act(W_dn [LayerNorm] x_in + b_dn) for an Adapter-like module with any of the reference's options
        (reference adapters.py:11-24).  Saves what the backward needs under sv[key] (t) and sv[key + "_pre"] (the
        pre-activation, for activations whose derivative is not a function of the output)."""
        from .adapters import activation_codes
        code, _, needs_pre = activation_codes(mod.act)
        xin = x_in if mod.ln is None else ops.layernorm(x_in, self._vec(mod.ln.weight), self._vec(mod.ln.bias), mod.ln.eps)
        if code == ops.MG_ACT_GELU_ERF:           # torch.nn.GELU(): its own pass, not an epilogue (ops.gelu_erf)
            pre = ops.gemm(xin, dn, layout="rm")
            t = ops.gelu_erf(pre)
        else:
            pre = torch.empty(x_in.shape[0], dn.N, dtype=BF16, device=x_in.device) if needs_pre else None
            t = ops.gemm(xin, dn, act=code, layout="rm", out2=pre)
        sv[key] = t
        if needs_pre:
            sv[key + "_pre"] = pre
        return t

    def _fgemm(self, key, x, lin, xq=None, mx_out=False, **kw):
        """GEMM against a FROZEN packed weight: bf16 tile GEMM, or (self.fp8) the fp8 MFMA on a per-row quantised x.
        fp8 only: ``mx_out`` -> the result exists only as an MX e4m3 operand ("mx", q, scales), written by the epilogue;
        an ``x`` of that form is multiplied by the MX-quantised weight (mg_gemm_mx_fp8), no quantisation pass."""
        if not self.fp8:
            return ops.gemm(x, lin, **kw)
        mx_in = isinstance(x, tuple)
        w8 = self._fp8_packs.get(key)
        if w8 is None or isinstance(w8, ops.PackedLinearMX) != mx_in:
            cls = ops.PackedLinearMX if mx_in else ops.PackedLinearFP8
            w8 = self._fp8_packs[key] = cls(ops.PackedLinear.untile(lin.ft)[: lin.N, : lin.K], lin.bias)
        # the MX copy is written by the 256x256 fp8 kernel (K % 256 == 0 in fp8 elements) for whole 32-column blocks; other
        # shapes (reduced test models) keep the bf16 output and the quantisation pass of the consumer
        mx_both = mx_out if isinstance(mx_out, tuple) else None      # (q, scales) from ops.mx_empty: the MX copy NEXT TO the bf16 output
        mx_ok = lin.K % 256 == 0 and lin.N % 32 == 0
        mx_out = bool(mx_out) and mx_both is None and mx_ok
        if mx_both is not None:
            assert mx_ok, "the MX output copy needs K % 256 == 0 and N % 32 == 0"
            kw.update(mx_out=mx_both, tile=256)
        elif mx_out:
            M = x[1].shape[0] if mx_in else (xq[0] if xq is not None else x).shape[0]
            kw.update(mx_out=ops.mx_empty(M, lin.N, self.device), no_out=True, tile=256)
        if mx_in:
            y = ops.gemm_mx_fp8(x[1], x[2], w8, **kw)
        else:
            q, sc = xq if xq is not None else ops.quantize_rows_fp8(x)
            y = ops.gemm_fp8(q, sc, w8, **kw)
        return ("mx", *kw["mx_out"]) if mx_out else y

    def forward_train(self, images, captions, dropout_mask=None, captions_host=None) -> LMOutput:
        model = self.module
        eng = model.lm.engine
        dev = self.device
        if captions_host is None:      # captions straight from the loader are host tensors; a device tensor costs one D2H sync
            captions_host = captions if not captions.is_cuda else captions.cpu()
        captions = captions.to(dev, non_blocking=True).contiguous()
        B, S_full = captions.shape
        assert S_full == model.seq_len, "captions must be padded to the sequence length (reference magma.py:249-251)"
        tape = {"B": B}
        # ---- image prefix (encoder tape inside) ----
        prefix, ptape = self._prefix_forward(images.to(dev), dropout_mask)
        tape["prefix"] = ptape
        P = prefix.shape[1]
        labels = ops.build_labels(captions, P, model.eos_token)
        S = S_full
        rows, tgt, first = self.target_index(captions_host, P, model.eos_token, S_full)
        if self.truncate:     # causal attention + masked loss => identical loss/grads
            S = min(S_full, ops.ceil_to(int(first.max()) + 1 + P + 1, 64))
            rows = (rows // S_full) * S + rows % S_full
        idx = torch.stack([rows, tgt]).pin_memory().to(dev, non_blocking=True)      # one small H2D copy, no sync
        tape["S"], tape["P"], tape["rows"], tape["tgt"] = S, P, idx[0], idx[1]
        emb = torch.empty(B, S, eng.d, dtype=BF16, device=dev)
        emb[:, :P] = prefix
        ops.embedding(captions[:, : S - P].contiguous(), eng.wte, emb, row_off=P)
        if self.lm_trainable:
            tape["caption_ids"] = captions
        loss = self._lm_forward(emb, labels[:, :S].contiguous(), tape)
        self._tape = tape
        # rows are b * S + position within the (possibly truncated) sequence of this step
        xf = tape.pop("x_final")
        # .logits is LAZY (language_model.LMOutput): reading it -- even `out.logits is not None` -- runs the (B*S x V) head GEMM,
        # and the closure keeps x_final [B*S, d] alive until the output is dropped; ask with out.is_lazy("logits").  With
        # truncate=True the tensor covers the positions this step ran, (B, logits_seq_len, V) with logits_seq_len <= seq_len (the
        # reference always returns (B, seq_len, V)); the rows cut off carry no target and were never computed.
        return LMOutput(loss=loss, labels=labels, target_rows=tape["rows"], target_logits=tape.pop("target_logits"),
                        logits_seq_len=S,
                        logits=LMOutput.lazy(lambda: eng._full_logits(xf, xf.shape[0]).view(B, S, eng.V)))

    def _lm_forward(self, emb, labels, tape):
        eng = self.module.lm.engine
        dev = self.device
        B, S, d = emb.shape
        M, H = B * S, eng.H
        x = emb.view(M, d)
        def block(li, ly, blk, x):
            """One GPT-J block forward: (x', what its backward needs).  Called by the loop below and -- under per-block
            recompute (self.recompute) -- once more per block from _lm_backward, on the saved block input."""
            sv = {"x": x}
            ln = ops.layernorm(x, ly.ln_g, ly.ln_b, eng.eps)
            lnq = ops.quantize_rows_fp8(ln) if self.fp8 else None     # shared by qkv and fc_in
            qkv = self._fgemm((li, "qkv"), ln, ly.qkv, lnq)
            a8 = rows = None
            if self.fp8 and self.fp8_attn:
                # e4m3 copies of q, k, v^T for the fp8 forward; the same pass writes the rotated q / k back into the qkv buffer (the
                # arithmetic of rotary_qk_inplace): the bf16 backward reads q, k, v as rows of that buffer -- no bf16 copies
                a8 = ops.rotary_split_fp8(qkv, B, S, H, eng.rot, eng.sin_t, eng.cos_t, inplace=True)
                rows = ops.AttnRows.of_qkv(qkv, B, S, H)
            elif os.environ.get("MAGMA_ATTN_TR", "1") == "0":
                # A/B switch only (tools/gpu_r06_step_ab.sh): the round-5 path -- split pass with three transposes, kernels with
                # transposed operand images
                vt_ld = ops.ceil_to(S, 32)
                q, k, v = (torch.empty(B, H, S, 256, dtype=BF16, device=dev) for _ in range(3))
                vt, qt, kt = (torch.empty(B, H, vt_ld // 32, 256, 32, dtype=BF16, device=dev) for _ in range(3))
                ops.rotary_split_train(qkv, B, S, H, eng.rot, eng.sin_t, eng.cos_t, q, k, v, vt, qt, kt)
                sv.update(old=(q, k, v, qt, kt))
            else:
                # round 6: no split pass and no transposed copies.  The rotary is applied in place to the q / k sections of the GEMM
                # output, and the attention kernels (csrc/attention_tr.hip) take q, k, v as strided rows of that buffer -- forward
                # and backward; the s-contraction fragments (V^T; Q^T, dO^T, K^T) come from the row images by ds_read_b64_tr_b16.
                ops.rotary_qk_inplace(qkv, B, S, H, eng.rot, eng.sin_t, eng.cos_t)
                rows = ops.AttnRows.of_qkv(qkv, B, S, H)
            out_up = self._cat_out_up(li, ly, blk)
            if out_up is not None:
                ctx_t = torch.empty(M, out_up.K, dtype=BF16, device=dev)        # [ctx | t]: one saved buffer, one GEMM operand
                ctx = ctx_t[:, :d]
            else:
                ctx = torch.empty(M, d, dtype=BF16, device=dev)
            lse = torch.empty(B, H, S, dtype=F32, device=dev)
            ctx_mx = None
            if a8 is not None:
                # fp8_mx: the attention epilogue also emits the OCP MX e4m3 copy of ctx (it holds whole rows): out_proj then runs as an
                # MX GEMM without a quantisation pass over ctx
                if self.fp8_mx and out_up is None and (H * 256) % 256 == 0:
                    ctx_mx = ops.mx_empty(M, H * 256, dev)
                ops.attn_prefill_fp8(a8, ctx, lse=lse, mx_out=ctx_mx)
            elif rows is None:
                ops.attn_prefill(q, k, vt, ctx, B, H, S, lse=lse)
            else:
                ops.attn_fwd_rows(rows, ctx, lse=lse)
            sv.update(rows=rows, ctx=ctx, lse=lse)
            a = None if out_up is not None else self._fgemm((li, "out"), ctx if ctx_mx is None else ("mx", *ctx_mx), ly.out)
            if ly.attn_adapter is not None and ly.attn_par is not None:
                # parallel / scaled_parallel (reference adapters.py:42-92): the adapter reads the attention INPUT (ln_1 output)
                dn, up, sc = self._par_adapter_ops(blk.attn)
                ta = self._adapter_down(blk.attn, ln, dn, sv, "ta")
                a = ops.gemm(ta, up, scale=sc, residuals=(a,), layout="rm")
            elif ly.attn_adapter is not None:
                dn, up = self._adapter_ops(blk.attn)
                ta = self._adapter_down(blk.attn, a, dn, sv, "ta")
                a2 = ops.gemm(ta, up, residuals=(a,), layout="rm")
                sv.update(a=a)
                a = a2
            hpre = torch.empty(M, ly.fc_in.N, dtype=BF16, device=dev)
            h = self._fgemm((li, "fc_in"), ln, ly.fc_in, lnq, act=ops.MG_ACT_GELU_NEW, out2=hpre, mx_out=self.fp8 and self.fp8_mx)
            sv["hpre"] = hpre
            if self.lm_trainable:
                sv["h"] = h                    # operand of the fc_out weight gradient
            if out_up is not None:
                # x' = [ctx | t] [W_out | W_up]^T + b_up + m + x   (reference adapters.py:38-39 on the MLP output + GPT-J's residual sum)
                dn, _ = self._adapter_ops(blk.mlp[1])
                m = self._fgemm((li, "fc_out"), h, ly.fc_out)
                sv["t"] = ops.gemm(m, dn, act=ops.MG_ACT_RELU, layout="rm", out=ctx_t[:, d:])
                x = ops.gemm(ctx_t, out_up, residuals=(m, x), layout="rm")
                sv.update(m=m)
            elif ly.mlp_adapter is not None and ly.mlp_par is not None:
                dn, up, sc = self._par_adapter_ops(blk.mlp)
                m = self._fgemm((li, "fc_out"), h, ly.fc_out)
                t = self._adapter_down(blk.mlp, ln, dn, sv, "t")
                x = ops.gemm(t, up, scale=sc, residuals=(m, a, x), layout="rm")
            elif self._ad8_ok(ly, blk.mlp[1]):
                # config[4]: the adapter GEMMs on the fp8 MFMA, operands from the producing epilogues (see __init__)
                p8 = self._ad8(li, blk.mlp[1])
                m_mx = ops.mx_empty(M, ly.fc_out.N, dev)
                m = self._fgemm((li, "fc_out"), h, ly.fc_out, mx_out=m_mx)
                t_mx = ops.mx_empty(M, p8["dn"].N, dev)
                t = ops.gemm_mx_fp8(m_mx[0], m_mx[1], p8["dn"], act=ops.MG_ACT_RELU, mx_out=t_mx, tile=256)
                x = ops.gemm_mx_fp8(t_mx[0], t_mx[1], p8["up"], residuals=(m, a, x))
                sv.update(m=m, t=t)
            elif ly.mlp_adapter is not None:
                dn, up = self._adapter_ops(blk.mlp[1])
                m = self._fgemm((li, "fc_out"), h, ly.fc_out)
                t = self._adapter_down(blk.mlp[1], m, dn, sv, "t")
                x = ops.gemm(t, up, residuals=(m, a, x), layout="rm")
                sv.update(m=m)
            else:
                x = self._fgemm((li, "fc_out"), h, ly.fc_out, residuals=(a, x))
            return x, sv

        # Per-block recompute (reference language_model.py:23-37: gradient checkpointing is ON by default there): only each
        # block's input is kept, its activations are rebuilt in the backward pass right before they are used (deterministic
        # kernels, no dropout inside the blocks: the same bits).  Off by default -- 288 GB hold the 176 GB of a B = 16 step,
        # and every reported number is taken without it; MAGMA_TRAIN_RECOMPUTE=1 / engine.recompute for larger
        # per-GPU batches or 384-pixel prefixes.
        saved = []
        for li, (ly, blk) in enumerate(zip(eng.layers, self.module.lm.transformer.h)):
            xin = x
            x, sv = block(li, ly, blk, x)
            saved.append({"x": xin} if self.recompute else sv)
        tape["block_fn"] = block if self.recompute else None
        tape["layers"] = saved
        # ---- head + loss on rows that carry a target ----
        rows, tgt = tape["rows"], tape["tgt"]             # built on the host (target_index): no device sync here
        if rows.numel() == 0:
            raise ValueError("no caption token carries a label in this batch")
        xr = x.index_select(0, rows)
        xl = ops.layernorm(xr, eng.lnf_g, eng.lnf_b, eng.eps)
        logits = torch.empty(xl.shape[0], eng.Vp, dtype=F32, device=dev)
        ops.gemm(xl, eng.head, out=logits)
        _, head_t = self._lm_packs()
        loss, dlogits = ops.cross_entropy_fwd_bwd(logits[:, : eng.V], tgt, head_t.K)
        tape.update(xr=xr, dlogits=dlogits, M=M)
        tape["target_logits"] = logits[:, : eng.V]         # fp32, rows that carry a target (reference magma.py:270-276 .logits, those rows)
        tape["x_final"] = x                                # [B*S, d]: .logits over every position is computed from it on first access
        if self.lm_trainable:
            tape["xl"] = xl
        return loss

    # ---- backward ----------------------------------------------------------------
    def _backward_impl(self, loss=None):
        tape = self._tape
        assert tape is not None, "backward() without a training forward"
        scale_note = 1.0 / self.gas     # applied at step() through grad_scale (DeepSpeed divides the loss by gas)
        del scale_note
        d_emb = self._lm_backward(tape)
        B, S, P = tape["B"], tape["S"], tape["P"]
        if isinstance(d_emb, tuple):      # the bottom block formed the gradient of the prefix rows only
            d_prefix = d_emb[1].view(B, P, -1)
        else:
            d_prefix = d_emb.view(B, S, -1)[:, :P].contiguous()
        if self.lm_trainable:       # word embedding rows of the caption tokens (positions P .. S-1)
            wte = self.module.lm.transformer.wte.weight
            ids = tape["caption_ids"][:, : S - P].reshape(-1)
            self.grad_of(wte).index_add_(0, ids, d_emb.view(B, S, -1)[:, P:].reshape(ids.shape[0], -1).float())
            self._reduce_params_async([wte, *self.module.lm.transformer.ln_f.parameters(), *self.module.lm.lm_head.parameters()])
        self._prefix_backward(tape["prefix"], d_prefix)
        self._tape = None
        self.micro_steps += 1

    def _acc_wgrad(self, param, gT: RawWeight, xT: RawWeight, row_scale=None):
        """grad(param)[N,K] += gT[N,M] . xT[K,M]^T  (fp32), optional per-row scale."""
        gview = self.grad_of(param).view(param.shape[0], -1)
        nk = gview.shape[1]
        if _WGRAD_INPLACE and nk % 4 == 0 and gview.data_ptr() % 16 == 0 and xT.N >= nk:
            # the GEMM's epilogue (or its split-K fix-up) adds row_scale[n] * (g^T x) into the fp32 gradient in place: no
            # temporary, no second pass (mg_epilogue.accumulate / .row_scale, ABI 6)
            xw = xT if xT.N == nk else RawWeight(xT.rm[:nk], K=xT.K)
            ops.gemm(gT.rm, xw, out=gview, layout="rm", use_bias=False, accumulate=True, row_scale=row_scale)
            return
        tmp = ops.gemm(gT.rm, xT, out_dtype=F32, layout="rm", use_bias=False)
        ops.scale_rows_acc(gview, tmp[:, : nk], row_scale)

    def _adapter_backward(self, mod, g, x_in, t, pre=None):
        """y = x_in + Wup act(Wdn [LN] x_in + bdn) + bup.  Given g = dL/dy accumulates the adapter's parameter gradients and
        returns (dt, Wdn^T): dL/d([LN] x_in) through the adapter is dt Wdn (see _adapter_dx for the way back to x_in)."""
        from .adapters import activation_codes
        dn, up = mod.down, mod.up
        _, gmode, needs_pre = activation_codes(mod.act)
        gT = RawWeight(ops.transpose_colsum(g, self.grad_of(up.bias)))         # g^T and d b_up from one pass over g
        self._acc_wgrad(up.weight, gT, _t(t))
        if gmode == ops.MG_AUX_GELU_ERF_GRAD:
            dt = ops.gemm(g, _t(up.weight.data), layout="rm", use_bias=False)
            ops.gelu_erf_grad_mul(dt, pre, out=dt)
        else:
            dt = ops.gemm(g, _t(up.weight.data), aux=pre if needs_pre else t, aux_mode=gmode, layout="rm", use_bias=False)
        dtT = RawWeight(ops.transpose_colsum(dt, self.grad_of(dn.bias)))      # dt^T and d b_dn
        xin = x_in if mod.ln is None else ops.layernorm(x_in, self._vec(mod.ln.weight), self._vec(mod.ln.bias), mod.ln.eps)
        self._acc_wgrad(dn.weight, dtT, _t(xin))
        return dt, _t(dn.weight.data)

    def _adapter_dx(self, mod, dt, dn_t, x_in, res=None):
        """dL/dx_in through the adapter's down-projection (+ res): dt Wdn, taken back through the adapter's LayerNorm (with its
        parameter gradients) when the adapter has one."""
        if mod.ln is None:
            return ops.gemm(dt, dn_t, residuals=() if res is None else (res,), layout="rm", use_bias=False)
        return self._ln_bwd(mod.ln, ops.gemm(dt, dn_t, layout="rm", use_bias=False), x_in, res=res)

    def _par_adapter_backward(self, wrapper, g, x_in, t, pre=None):
        """y = f(x_in) + s * (Wup relu(Wdn x_in + bdn) + bup)  (reference adapters.py:59-63, :84-91).  Given g = dL/dy:
        accumulates the adapter's parameter gradients (and ds = <g, z>, z the unscaled adapter output, computed from the
        unscaled weight / bias gradients: <g^T t, Wup> + <colsum g, bup>) and returns (dt, Wdn^T): dL/dx_in through the
        adapter is dt Wdn, added by the caller."""
        from .adapters import activation_codes
        dn, up = wrapper.down, wrapper.up
        _, gmode, needs_pre = activation_codes(wrapper.act)
        sp = wrapper.adapter_scale
        scaled = torch.is_tensor(sp)
        N = up.weight.shape[0]
        sc = None
        if scaled:
            sval = self.master_of(sp) if self.is_trainable(sp) else sp.detach().float()
            sc = sval.reshape(1).to(F32).expand(N).contiguous()
        gb = torch.zeros(N, dtype=F32, device=g.device)
        ops.colsum(g, gb)                                                  # unscaled d b_up
        gw = ops.gemm(_t(g).rm, _t(t), out_dtype=F32, layout="rm", use_bias=False)     # unscaled d W_up [N, K]
        gview = self.grad_of(up.weight).view(N, -1)
        ops.scale_rows_acc(gview, gw[:, : gview.shape[1]], sc)
        self.grad_of(up.bias).add_(gb * sc if scaled else gb)
        if scaled and self.is_trainable(sp):
            ds = (gw[:, : gview.shape[1]] * up.weight.data.float()).sum() + (gb * self.master_of(up.bias)).sum()
            self.grad_of(sp).add_(ds.reshape(self.grad_of(sp).shape))
        K = up.weight.shape[1]
        if gmode == ops.MG_AUX_GELU_ERF_GRAD:
            dt = ops.gemm(g, _t(up.weight.data), layout="rm", use_bias=False, scale=None if sc is None else sc[:K].contiguous())
            ops.gelu_erf_grad_mul(dt, pre, out=dt)
        else:
            dt = ops.gemm(g, _t(up.weight.data), aux=pre if needs_pre else t, aux_mode=gmode, layout="rm", use_bias=False,
                          scale=None if sc is None else sc[:K].contiguous())
        ops.colsum(dt, self.grad_of(dn.bias))
        xin = x_in if wrapper.ln is None else ops.layernorm(x_in, self._vec(wrapper.ln.weight), self._vec(wrapper.ln.bias), wrapper.ln.eps)
        self._acc_wgrad(dn.weight, _t(dt), _t(xin))
        return dt, _t(dn.weight.data)

    def _lm_backward(self, tape):
        eng = self.module.lm.engine
        dev = self.device
        packs, head_t = self._lm_packs()
        B, S, M = tape["B"], tape["S"], tape["M"]
        H, d = eng.H, eng.d
        # loss head: dlogits -> dxl -> ln_f backward -> scatter to the target rows
        lm = self.module.lm
        dxl = ops.gemm(tape["dlogits"], head_t)
        if self.lm_trainable:
            V = lm.lm_head.weight.shape[0]
            dlT = ops.transpose(tape["dlogits"])                       # [Vp, R]
            self._acc_wgrad(lm.lm_head.weight, RawWeight(dlT[:V]), _t(tape["xl"]))
            if lm.lm_head.bias is not None:
                cs = torch.zeros(tape["dlogits"].shape[1], dtype=F32, device=dev)
                ops.colsum(tape["dlogits"], cs)
                self.grad_of(lm.lm_head.bias).add_(cs[:V])
            dxr = self._ln_bwd(lm.transformer.ln_f, dxl, tape["xr"])
        else:
            dxr = ops.layernorm_bwd(dxl, tape["xr"], eng.lnf_g, eng.eps)
        g = torch.zeros(M, d, dtype=BF16, device=dev)
        g.index_copy_(0, tape["rows"], dxr)
        for li in range(len(eng.layers) - 1, -1, -1):
            ly, blk, pk, sv = eng.layers[li], self.module.lm.transformer.h[li], packs[li], tape["layers"][li]
            if tape.get("block_fn") is not None:       # per-block recompute: rebuild this block's activations from its input
                _, sv = tape["block_fn"](li, ly, blk, sv["x"])
                tape["layers"][li] = None
            # The BOTTOM block of a frozen LM: below it only the image prefix (positions < P of every sequence) receives a gradient
            # -- the word embeddings are frozen (reference magma.py:98-100 trains adapters + image prefix) --, so the three dgrads
            # into ln_1 (fc_out^T, fc_in^T, qkv^T), dQ and the LayerNorm backward are needed for B*P of the B*S rows only, and dK / dV
            # for the first P keys.  Every parameter gradient is what it was (the adapter branch runs on all rows).
            bottom = (_BOTTOM_PREFIX_ONLY and li == 0 and not self.lm_trainable and 0 < tape["P"] < S and sv.get("rows") is not None
                      and ly.mlp_par is None and ly.attn_par is None)
            P = tape["P"]
            take = (lambda t_: t_.view(B, S, -1)[:, :P].reshape(B * P, -1)) if bottom else (lambda t_: t_)
            if li == 0:
                self.bottom_prefix_rows = P if bottom else 0      # (bench.py: executed-FLOP accounting)
            # ---- MLP branch ----
            par = ly.mlp_par is not None or ly.attn_par is not None
            ln = ops.layernorm(sv["x"], ly.ln_g, ly.ln_b, eng.eps) if par else None   # the parallel adapters' input, recomputed
            extra = []                                                                 # dL/d ln through the parallel adapters
            gq = None                                                                  # row-quantised g, shared by its consumers
            dm_taken = False                                                           # bottom block: dm already holds the prefix rows only
            if ly.mlp_adapter is not None and ly.mlp_par is not None:
                dt, dn_t = self._par_adapter_backward(blk.mlp, g, ln, sv["t"], sv.get("t_pre"))
                extra.append(self._adapter_dx(blk.mlp, dt, dn_t, ln))
                dm = g
            elif self._ad8_ok(ly, blk.mlp[1]) and not bottom:
                mod, p8 = blk.mlp[1], self._ad8(li, blk.mlp[1])
                gq = ops.quantize_rows_fp8(g)          # also the operand of the out_proj dgrad below (no attention adapter: da = g)
                self._acc_wgrad(mod.up.weight, RawWeight(ops.transpose_colsum(g, self.grad_of(mod.up.bias))), _t(sv["t"]))
                dt_mx = ops.mx_empty(M, p8["up_t"].N, dev)
                dt = ops.gemm_fp8(gq[0], gq[1], p8["up_t"], aux=sv["t"], aux_mode=ops.MG_AUX_RELU_GATE, use_bias=False,
                                  mx_out=dt_mx, tile=256)
                self._acc_wgrad(mod.down.weight, RawWeight(ops.transpose_colsum(dt, self.grad_of(mod.down.bias))), _t(sv["m"]))
                dm_mx = ops.mx_empty(M, p8["dn_t"].N, dev)
                ops.gemm_mx_fp8(dt_mx[0], dt_mx[1], p8["dn_t"], residuals=(g,), use_bias=False, mx_out=dm_mx, no_out=True, tile=256)
                dm = ("mx", *dm_mx)                    # dL/dm exists only as the MX operand of the fc_out dgrad
            elif ly.mlp_adapter is not None:
                dt, dn_t = self._adapter_backward(blk.mlp[1], g, sv["m"], sv["t"], sv.get("t_pre"))
                if bottom and blk.mlp[1].ln is None:
                    # dL/dm feeds nothing but the (prefix-rows-only) fc_out dgrad: the same rows suffice here
                    dm = self._adapter_dx(blk.mlp[1], take(dt), dn_t, None, res=take(g))
                    dm_taken = True
                else:
                    dm = self._adapter_dx(blk.mlp[1], dt, dn_t, sv["m"], res=g)
            else:
                dm = g
            dhpre = self._fgemm((li, "fc_out_t"), dm if dm_taken else take(dm), pk["fc_out_t"], aux=take(sv["hpre"]), aux_mode=ops.MG_AUX_GELU_GRAD,
                                mx_out=self.fp8 and self.fp8_mx and not bottom)
            dln_mlp = self._fgemm((li, "fc_in_t"), dhpre, pk["fc_in_t"])
            if self.lm_trainable:
                a_mod, mlp_mod = ly._src
                if ln is None:
                    ln = ops.layernorm(sv["x"], ly.ln_g, ly.ln_b, eng.eps)
                ops.colsum(dm, self.grad_of(mlp_mod.c_proj.bias))
                self._acc_wgrad(mlp_mod.c_proj.weight, _t(dm), _t(sv["h"]))
                ops.colsum(dhpre, self.grad_of(mlp_mod.c_fc.bias))
                self._acc_wgrad(mlp_mod.c_fc.weight, _t(dhpre), _t(ln))
            del dhpre, dm
            # ---- attention branch ----
            if ly.attn_adapter is not None and ly.attn_par is not None:
                dta, dn_t = self._par_adapter_backward(blk.attn, g, ln, sv["ta"], sv.get("ta_pre"))
                extra.append(self._adapter_dx(blk.attn, dta, dn_t, ln))
                da = g
            elif ly.attn_adapter is not None:
                dta, dn_t = self._adapter_backward(blk.attn, g, sv["a"], sv["ta"], sv.get("ta_pre"))
                da = self._adapter_dx(blk.attn, dta, dn_t, sv["a"], res=g)
            else:
                da = g
            dctx = self._fgemm((li, "out_t"), da, pk["out_t"], xq=gq if da is g else None)
            if sv["rows"] is None:      # MAGMA_ATTN_TR=0 (A/B only)
                q, k, v, qt, kt = sv["old"]
                dqkv = ops.attn_bwd_merged(q, k, v, qt, kt, dctx, sv["ctx"], sv["lse"], B, H, S, eng.rot, eng.sin_t, eng.cos_t)
            elif bottom:
                # dQ of the first ceil(P / 128) query blocks, dK / dV of the first key blocks (mg_attn_bwd_rows_bf16 first_rows)
                dqkv = take(ops.attn_bwd_rows(sv["rows"], dctx, sv["ctx"], sv["lse"], merged_rot=(eng.rot, eng.sin_t, eng.cos_t), first_rows=P))
            elif self.fp8 and self.fp8_mx and (3 * H * 256) % 256 == 0:
                # fp8_mx: the gradient of the fused qkv projection leaves the attention backward's epilogues ONLY as the OCP MX e4m3
                # operand of the qkv dgrad (the LM is frozen in fp8 mode: nothing else reads dqkv) -- no bf16 dqkv, no quantisation pass
                dq_mx = ops.mx_empty(M, 3 * H * 256, dev)
                ops.attn_bwd_rows(sv["rows"], dctx, sv["ctx"], sv["lse"], merged_rot=(eng.rot, eng.sin_t, eng.cos_t), mx_out=dq_mx, no_out=True)
                dqkv = ("mx", *dq_mx)
            else:
                dqkv = ops.attn_bwd_rows(sv["rows"], dctx, sv["ctx"], sv["lse"], merged_rot=(eng.rot, eng.sin_t, eng.cos_t))
            dln = self._fgemm((li, "qkv_t"), dqkv, pk["qkv_t"], residuals=(dln_mlp, *extra))
            if self.lm_trainable:
                self._acc_wgrad(a_mod.out_proj.weight, _t(da), _t(sv["ctx"]))
                dqkvT, lnT, dd = ops.transpose(dqkv), _t(ln), d
                for i3, prj in enumerate((a_mod.q_proj, a_mod.k_proj, a_mod.v_proj)):
                    self._acc_wgrad(prj.weight, RawWeight(dqkvT[i3 * dd:(i3 + 1) * dd]), lnT)
                g = self._ln_bwd(blk.ln_1, dln, sv["x"], res=g)
            else:
                g = ops.layernorm_bwd(dln, take(sv["x"]), ly.ln_g, eng.eps, res=take(g))
                if bottom:
                    g = ("prefix", g)     # [B * P, d]: the gradient of the image prefix; the other rows were never formed
            tape["layers"][li] = None     # free this layer's activations
            self._reduce_params_async([p for p in blk.parameters() if p.requires_grad])
        return g

    # ---- image prefix + CLIP trunk -------------------------------------------------
    def _prefix_forward(self, images, dropout_mask):
        ip = self.module.image_prefix
        if ip.pooled:
            return self._pooled_prefix_forward(images, dropout_mask)
        feats, etape = self._encoder_forward(images)
        B, P, E = feats.shape
        proj = RawWeight(ip.proj.weight.data, bias=self.master_of(ip.proj.bias)) if self.is_trainable(ip.proj.weight) \
            else PackedLinear(ip.proj.weight, ip.proj.bias)
        f2 = feats.reshape(B * P, E)
        keep = 1.0 - ip.dropout.p
        mask = None
        if ip.dropout.p > 0:
            if dropout_mask is None:
                dropout_mask = (torch.rand(B * P, ip.out_dim, device=self.device) < keep).to(BF16) / keep
            mask = dropout_mask.reshape(B * P, ip.out_dim).to(BF16).contiguous()
        y1 = ops.gemm(f2, proj, layout="rm" if isinstance(proj, RawWeight) else None,
                      aux=mask, aux_mode=ops.MG_AUX_MUL if mask is not None else ops.MG_AUX_NONE)
        if ip.use_layernorm:
            y2 = ops.layernorm(y1, self.master_of(ip.ln.weight), self.master_of(ip.ln.bias), ip.ln.eps)
        else:
            y2 = y1
        return y2.view(B, P, ip.out_dim), {"enc": etape, "feats": f2, "mask": mask, "y1": y1, "B": B, "P": P}

    def _prefix_backward(self, pt, d_prefix):
        ip = self.module.image_prefix
        if ip.pooled:
            return self._pooled_prefix_backward(pt, d_prefix)
        B, P = pt["B"], pt["P"]
        g = d_prefix.reshape(B * P, ip.out_dim)
        if ip.use_layernorm:
            dx, xhat = ops.layernorm_bwd(g, pt["y1"], self.master_of(ip.ln.weight), ip.ln.eps, want_xhat=True)
            ops.colsum(g, self.grad_of(ip.ln.weight), xhat)
            ops.colsum(g, self.grad_of(ip.ln.bias))
            g = dx
        if pt["mask"] is not None:
            g = ops.mul(g, pt["mask"])
        ops.colsum(g, self.grad_of(ip.proj.bias))
        self._acc_wgrad(ip.proj.weight, _t(g), _t(pt["feats"]))
        self._reduce_params_async([p for n, p in ip.named_parameters() if not n.startswith("enc.")])
        if pt["enc"] is None:
            return
        # gradient wrt the trunk output (a post-ReLU tensor): gate fused in the epilogue
        d_feats = ops.gemm(g, _t(ip.proj.weight.data), aux=pt["feats"], aux_mode=ops.MG_AUX_RELU_GATE, layout="rm",
                           use_bias=False)
        self._encoder_backward(pt["enc"], d_feats)

    # ---- pooled prefix over the CLIP ViT (encoder_name "clip"; reference image_prefix.py:60-72, 85-109) ------------
    def _lin(self, weight, bias):
        """GEMM operand on a live [N, K] parameter (fp32 master bias), or the frozen tensors when it is not trained."""
        if self.is_trainable(weight):
            return RawWeight(weight.data, bias=None if bias is None else self.master_of(bias))
        return RawWeight(weight.data, bias=None if bias is None else bias.detach().float().contiguous())

    def _vec(self, p):
        return self.master_of(p) if self.is_trainable(p) else p.detach().float().contiguous()

    def _ln_bwd(self, ln, dy, x, res=None):
        """LayerNorm backward with its parameter gradients: returns dL/dx (+ res)."""
        if self.is_trainable(ln.weight):
            dx, xhat = ops.layernorm_bwd(dy, x, self._vec(ln.weight), ln.eps, res=res, want_xhat=True)
            ops.colsum(dy, self.grad_of(ln.weight), xhat)
            ops.colsum(dy, self.grad_of(ln.bias))
            return dx
        return ops.layernorm_bwd(dy, x, self._vec(ln.weight), ln.eps, res=res)

    def _lin_bwd(self, weight, bias, g, x_in, want_dx=True, **kw):
        """y = x_in W^T + b: accumulates dW, db and returns dL/dx_in = g W (epilogue options in kw)."""
        if self.is_trainable(weight):
            if bias is not None:
                ops.colsum(g, self.grad_of(bias))
            self._acc_wgrad(weight, _t(g), _t(x_in))
        if not want_dx:
            return None
        return ops.gemm(g, _t(weight.data.view(weight.shape[0], -1)), layout="rm", use_bias=False, **kw)

    def _pooled_encoder_forward(self, images):
        from .image_encoders import NFResNet50
        enc = self.module.image_prefix.enc
        if not any(self.is_trainable(p) for p in enc.parameters()):
            return enc(images), None               # frozen encoder: inference path, nothing taped
        if isinstance(enc, NFResNet50):
            return self._nf_forward(images)
        return self._vit_forward(images)

    # ---- timm NF-ResNet-50 (encoder_name "nfresnet50"; reference image_encoders.py:31-45), trained -------------------------
    def _nf_scale(self, c: int, v: float):
        key = (c, round(v, 9))
        t = self._nf_scales.get(key)
        if t is None:
            t = self._nf_scales[key] = torch.full((c,), v, dtype=F32, device=self.device)
        return t

    def _nf_conv_fwd(self, conv, a, mult=1.0, bias_mult=1.0, relu=False, geom=None, residual=None, stem=False):
        """y = act(mult * (W_hat a) + bias_mult * b) (+ residual) with W_hat = the scaled-standardised LIVE weights; returns
        (y, record for the backward)."""
        from .image_encoders import NFResNet50 as NF
        w = conv.weight.data
        cout, cin, k, _ = w.shape
        sc = NF.GAMMA * (cin * k * k) ** -0.5
        wh = ops.weight_standardize(w.contiguous(), conv.gain.data.reshape(-1).contiguous(), sc, NF.EPS,
                                    ldo=160 if stem else None)                        # [cout, fan_in] in (cin, ky, kx) order
        bias = self.master_of(conv.bias) if self.is_trainable(conv.bias) else conv.bias.detach().float()
        if bias_mult != 1.0:
            bias = bias * bias_mult
        convarg = None
        if stem or k == 1:
            wop = RawWeight(wh, bias=bias)
        else:
            wop = RawWeight(ops.conv_weight_relayout(wh.view(cout, cin, k, k), 0), bias=bias, K=k * k * cin)
            convarg = (geom[1], geom[2], cin)
        y = ops.gemm(a, wop, scale=self._nf_scale(cout, mult), act=ops.MG_ACT_RELU if relu else ops.MG_ACT_NONE, conv=convarg,
                     residuals=() if residual is None else (residual,), layout="rm")
        return y, {"conv": conv, "a": a, "wh": wh, "sc": sc, "mult": mult, "bias_mult": bias_mult, "geom": geom,
                   "kind": "stem" if stem else ("1x1" if k == 1 else "3x3")}

    def _nf_conv_bwd(self, rec, g, need_dgrad=True, gate=None, residuals=()):
        """g = gradient wrt the conv's output y (ReLU already gated by the caller).  Accumulates bias / weight / gain gradients
        through the weight standardisation; returns dL/da, zeroed where ``gate`` <= 0, plus ``residuals``."""
        from .image_encoders import NFResNet50 as NF
        conv, a, wh = rec["conv"], rec["a"], rec["wh"]
        cout, cin, k, _ = conv.weight.shape
        if self.is_trainable(conv.bias):
            if rec["bias_mult"] == 1.0:
                ops.colsum(g, self.grad_of(conv.bias))
            else:
                tmpb = torch.zeros(cout, dtype=F32, device=g.device)
                ops.colsum(g, tmpb)
                self.grad_of(conv.bias).add_(tmpb, alpha=rec["bias_mult"])
        if self.is_trainable(conv.weight):
            gT = _t(g)
            if rec["kind"] == "3x3":
                Bq, hh, ww = rec["geom"]
                xT = RawWeight(ops.im2col_t(a, Bq, hh, ww, cin))
            else:
                xT = _t(a)
            dwh = ops.gemm(gT.rm, xT, out_dtype=F32, layout="rm", use_bias=False)       # [cout, >= fan_in], weight's own order
            ops.weight_standardize_bwd(conv.weight.data.contiguous(), conv.gain.data.reshape(-1).contiguous(), dwh,
                                       self.grad_of(conv.weight).view(cout, -1), self.grad_of(conv.gain).view(-1),
                                       rec["sc"], NF.EPS, dmult=rec["mult"])
        if not need_dgrad:
            return None
        aux_kw = dict(aux=gate, aux_mode=ops.MG_AUX_RELU_GATE) if gate is not None else {}
        fan_in = cin * k * k
        w4 = wh[:, :fan_in].contiguous().view(cout, cin, k, k)
        wop = RawWeight(ops.conv_weight_relayout(w4, 1, self._nf_scale(cout, rec["mult"])), K=k * k * cout)
        if rec["kind"] == "3x3":
            Bq, hh, ww = rec["geom"]
            return ops.gemm(g, wop, conv=(hh, ww, cout), layout="rm", use_bias=False, residuals=residuals, **aux_kw)
        return ops.gemm(g, wop, layout="rm", use_bias=False, residuals=residuals, **aux_kw)

    def _nf_forward(self, images):
        enc = self.module.image_prefix.enc
        x = images.to(BF16).contiguous()
        B, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError(f"nfresnet50 takes images whose sides are multiples of 32, got {H}x{W}")
        h, w = H // 2, W // 2
        cols = ops.im2col_nchw(x, 7, 2, 3, 160)
        y0, stem = self._nf_conv_fwd(enc.stem_conv, cols, stem=True)                      # [B*h*w, 64], no activation
        y = ops.maxpool3x3s2(y0.view(B, h, w, -1))
        tape = {"nf": True, "stem": stem, "y0": y0, "geom0": (B, h, w), "blocks": []}
        h, w = y.shape[1], y.shape[2]
        y = y.view(B * h * w, -1)
        one, zero = self._nf_scale(enc.out_dim, 1.0), self._nf_scale(enc.out_dim, 0.0)
        for stage in enc.stages:
            for blk in stage:
                c = y.shape[1]
                r = ops.bn_apply(y, one[:c], zero[:c], relu=True)                        # relu(x); beta rides in the epilogues
                br = {"x": y, "geom": (B, h, w), "stride": blk.stride}
                shortcut = y
                if blk.downsample is not None:
                    s_in = ops.avgpool2(r.view(B, h, w, c)).view(B * (h // 2) * (w // 2), c) if blk.stride > 1 else r
                    shortcut, br["ud"] = self._nf_conv_fwd(blk.downsample.conv, s_in, mult=blk.beta)
                o1, br["u1"] = self._nf_conv_fwd(blk.conv1, r, mult=blk.beta, relu=True)
                o2, br["u2"] = self._nf_conv_fwd(blk.conv2, o1, relu=True, geom=(B, h, w))
                br["o1"], br["o2"] = o1, o2
                if blk.stride > 1:
                    o2 = ops.subsample2(o2.view(B, h, w, -1))
                    h, w = o2.shape[1], o2.shape[2]
                    o2 = o2.view(B * h * w, -1)
                y, br["u3"] = self._nf_conv_fwd(blk.conv3, o2, mult=enc.ALPHA, bias_mult=enc.ALPHA, residual=shortcut)
                br["out_geom"] = (B, h, w)
                tape["blocks"].append(br)
        tape["y_last"] = y.view(B, h * w, -1)
        return ops.relu_mean_rows(tape["y_last"]), tape

    def _nf_backward(self, tape, d_feats):
        """d_feats: gradient wrt the pooled (B, 2048) features."""
        enc = self.module.image_prefix.enc
        yl = tape["y_last"]
        B = yl.shape[0]
        g = ops.relu_mean_rows_bwd(yl, d_feats.contiguous()).view(-1, yl.shape[2])       # dL/dy of the last block
        blocks = [blk for stage in enc.stages for blk in stage]
        for blk, br in zip(reversed(blocks), reversed(tape["blocks"])):
            Bq, h, w = br["geom"]
            _, ho, wo = br["out_geom"]
            x = br["x"]
            # residual branch: y = alpha * conv3(o2s) + shortcut
            g2 = self._nf_conv_bwd(br["u3"], g)                                           # wrt o2 (sub-sampled), ungated
            if br["stride"] > 1:
                g2 = ops.subsample2_bwd(g2.view(Bq, ho, wo, -1), h, w).view(Bq * h * w, -1)
            g2 = ops.add_gate(g2.contiguous(), gate=br["o2"])                              # ReLU of conv2's output
            g1 = self._nf_conv_bwd(br["u2"], g2, gate=br["o1"])                           # wrt o1, gated by its ReLU
            d_r = self._nf_conv_bwd(br["u1"], g1)                                         # wrt r = relu(x)
            # shortcut branch
            if "ud" in br:
                d_s = self._nf_conv_bwd(br["ud"], g)
                if br["stride"] > 1:
                    d_s = ops.avgpool2_bwd(d_s.view(Bq, ho, wo, -1), Bq, h, w, d_s.shape[1]).view(Bq * h * w, -1)
                g = ops.add_gate(d_r, d_s, gate=x)                                        # both paths go through relu(x)
            else:
                g = ops.add_gate(ops.add_gate(d_r, gate=x), g)                            # relu path gated, identity path not
        B0, h0, w0 = tape["geom0"]
        g0 = ops.maxpool3x3s2_bwd(tape["y0"].view(B0, h0, w0, -1), g.view(B0, (h0 - 1) // 2 + 1, (w0 - 1) // 2 + 1, -1))
        self._nf_conv_bwd(tape["stem"], g0.view(B0 * h0 * w0, -1), need_dgrad=False)
        self._reduce_params_async([p for p in enc.parameters() if self.is_trainable(p)])

    def _vit_forward(self, images):
        """Taped forward of image_encoders.VisionTransformer.forward (same kernels, live parameters)."""
        enc = self.module.image_prefix.enc
        Pz, w, Hh = enc.patch_size, enc.width, enc.heads
        x = images.to(BF16).contiguous()
        B = x.shape[0]
        patches = ops.patchify(x, Pz)                                           # [B*G, 3*P*P]
        pe = ops.gemm(patches, RawWeight(enc.conv1.weight.data.view(w, -1)), layout="rm", use_bias=False)
        emb = ops.vit_embed(pe, enc.class_embedding.data, enc.positional_embedding.data, B)
        S = emb.shape[1]
        emb = emb.view(B * S, w)
        t = ops.layernorm(emb, self._vec(enc.ln_pre.weight), self._vec(enc.ln_pre.bias), enc.ln_pre.eps)
        blocks = []
        for blk in enc.transformer.resblocks:
            h1 = ops.layernorm(t, self._vec(blk.ln_1.weight), self._vec(blk.ln_1.bias), blk.ln_1.eps)
            qkv = ops.gemm(h1, self._lin(blk.attn.in_proj_weight, blk.attn.in_proj_bias), layout="rm")
            ctx = ops.attn_small(qkv, B, S, Hh)
            tm = ops.gemm(ctx, self._lin(blk.attn.out_proj.weight, blk.attn.out_proj.bias), residuals=(t,), layout="rm")
            h2 = ops.layernorm(tm, self._vec(blk.ln_2.weight), self._vec(blk.ln_2.bias), blk.ln_2.eps)
            fpre = torch.empty(B * S, 4 * w, dtype=BF16, device=self.device)
            f = ops.gemm(h2, self._lin(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias), act=ops.MG_ACT_QUICK_GELU, out2=fpre, layout="rm")
            tn = ops.gemm(f, self._lin(blk.mlp.c_proj.weight, blk.mlp.c_proj.bias), residuals=(tm,), layout="rm")
            blocks.append({"t": t, "h1": h1, "qkv": qkv, "ctx": ctx, "tm": tm, "h2": h2, "fpre": fpre, "f": f})
            t = tn
        cls = t.view(B, S, w)[:, 0, :].contiguous()
        cln = ops.layernorm(cls, self._vec(enc.ln_post.weight), self._vec(enc.ln_post.bias), enc.ln_post.eps)
        out = ops.gemm(cln, RawWeight(ops.transpose(enc.proj.data)), layout="rm", use_bias=False)     # (B, out_dim)
        return out, {"patches": patches, "emb": emb, "blocks": blocks, "cls": cls, "cln": cln, "B": B, "S": S}

    def _vit_backward(self, vt, d_out):
        enc = self.module.image_prefix.enc
        B, S, w, Hh = vt["B"], vt["S"], enc.width, enc.heads
        # out = cln @ proj  (proj is stored [width, out_dim])
        if self.is_trainable(enc.proj):
            self._acc_wgrad(enc.proj, _t(vt["cln"]), _t(d_out))
        d_cln = ops.gemm(d_out, RawWeight(enc.proj.data), layout="rm", use_bias=False)                # [B, width]
        d_cls = self._ln_bwd(enc.ln_post, d_cln, vt["cls"])
        g = torch.zeros(B, S, w, dtype=BF16, device=self.device)
        g[:, 0, :] = d_cls
        g = g.view(B * S, w)
        for blk, sv in zip(reversed(list(enc.transformer.resblocks)), reversed(vt["blocks"])):
            # tn = tm + c_proj(QuickGELU(c_fc(ln_2 tm)))
            d_fpre = self._lin_bwd(blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, g, sv["f"], aux=sv["fpre"],
                                   aux_mode=ops.MG_AUX_QUICK_GELU_GRAD)
            d_h2 = self._lin_bwd(blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, d_fpre, sv["h2"])
            gm = self._ln_bwd(blk.ln_2, d_h2, sv["tm"], res=g)
            # tm = t + out_proj(attn(in_proj(ln_1 t)))
            d_ctx = self._lin_bwd(blk.attn.out_proj.weight, blk.attn.out_proj.bias, gm, sv["ctx"])
            d_qkv = ops.attn_small_bwd(sv["qkv"], d_ctx, B, S, Hh)
            d_h1 = self._lin_bwd(blk.attn.in_proj_weight, blk.attn.in_proj_bias, d_qkv, sv["h1"])
            g = self._ln_bwd(blk.ln_1, d_h1, sv["t"], res=gm)
            self._reduce_params_async([p for p in blk.parameters() if self.is_trainable(p)])
        d_emb = self._ln_bwd(enc.ln_pre, g, vt["emb"]).view(B, S, w)
        if self.is_trainable(enc.positional_embedding):
            self.grad_of(enc.positional_embedding).add_(d_emb.float().sum(0))
        if self.is_trainable(enc.class_embedding):
            self.grad_of(enc.class_embedding).add_(d_emb[:, 0, :].float().sum(0))
        if self.is_trainable(enc.conv1.weight):
            d_pe = d_emb[:, 1:, :].reshape(B * (S - 1), w).contiguous()
            self._acc_wgrad(enc.conv1.weight, _t(d_pe), _t(vt["patches"]))
        self._reduce_params_async([p for p in enc.parameters() if self.is_trainable(p)])

    def _pooled_prefix_forward(self, images, dropout_mask):
        ip = self.module.image_prefix
        feats, vtape = self._pooled_encoder_forward(images)                      # (B, enc_dim)
        B, s, d = feats.shape[0], ip.out_seq_len, ip.out_dim
        y0 = ops.gemm(feats.contiguous(), self._lin(ip.proj.weight, ip.proj.bias), layout="rm").reshape(B * s, d)
        mask = None
        if ip.dropout.p > 0:
            if dropout_mask is None:
                keep = 1.0 - ip.dropout.p
                dropout_mask = (torch.rand(B * s, d, device=self.device) < keep).to(BF16) / keep
            mask = dropout_mask.reshape(B * s, d).to(BF16).contiguous()
            y1 = ops.mul(y0.contiguous(), mask)
        else:
            y1 = y0.contiguous()
        y2 = ops.layernorm(y1, self._vec(ip.ln.weight), self._vec(ip.ln.bias), ip.ln.eps) if ip.use_layernorm else y1
        return y2.view(B, s, d), {"vit": vtape, "feats": feats, "mask": mask, "y1": y1, "B": B}

    def _pooled_prefix_backward(self, pt, d_prefix):
        ip = self.module.image_prefix
        B, s, d = pt["B"], ip.out_seq_len, ip.out_dim
        g = d_prefix.reshape(B * s, d).contiguous()
        if ip.use_layernorm:
            g = self._ln_bwd(ip.ln, g, pt["y1"])
        if pt["mask"] is not None:
            g = ops.mul(g, pt["mask"])
        g = g.view(B, s * d)
        d_feats = self._lin_bwd(ip.proj.weight, ip.proj.bias, g, pt["feats"], want_dx=pt["vit"] is not None)
        self._reduce_params_async([p for n, p in ip.named_parameters() if not n.startswith("enc.")])
        if pt["vit"] is not None:
            if "nf" in pt["vit"]:
                self._nf_backward(pt["vit"], d_feats)
            else:
                self._vit_backward(pt["vit"], d_feats)

    # The trunk is executed unit by unit; a unit = conv (+ frozen-statistics BN) (+ ReLU).
    def _bn_vectors(self, bn):
        gamma, beta = self.master_of(bn.weight), self.master_of(bn.bias)
        st = self._bn_stats.get(id(bn))
        if st is None:      # frozen statistics: fp32 copies made once (dropped when a checkpoint is loaded)
            st = self._bn_stats[id(bn)] = (bn.running_mean.float().contiguous(), bn.running_var.float().contiguous())
        scale, shift = ops.bn_fold(gamma, beta, st[0], st[1], bn.eps)
        return gamma, beta, scale, shift

    def _conv_plan(self, enc):
        """Frozen-statistics mode: the GEMM operands of every trunk convolution (forward and dgrad layout) and the folded affine of
        its BatchNorm live in ONE buffer each and are re-derived from the current weights by two launches per forward
        (ops.ConvOperandPlan) -- before: one re-layout per convolution and direction plus one fold per BatchNorm, ~380 launches."""
        if self._plan is None:
            units, idx = [], {}
            pairs = [(enc.conv1, enc.bn1), (enc.conv2, enc.bn2), (enc.conv3, enc.bn3)]      # the pairs _encoder_forward walks
            for li in range(1, 5):
                for blk in getattr(enc, f"layer{li}"):
                    pairs += [(blk.conv1, blk.bn1), (blk.conv2, blk.bn2), (blk.conv3, blk.bn3)]
                    if blk.downsample is not None:
                        pairs.append((blk.downsample[1], blk.downsample[2]))
            for conv, bn in pairs:
                st = self._bn_stats.get(id(bn))
                if st is None:
                    st = self._bn_stats[id(bn)] = (bn.running_mean.float().contiguous(), bn.running_var.float().contiguous())
                idx[id(conv)] = len(units)
                units.append((conv.weight.data, self.master_of(bn.weight), self.master_of(bn.bias), st[0], st[1], bn.eps))
            self._plan = (ops.ConvOperandPlan(units, self.device), idx)
        return self._plan

    def _unit_fwd(self, conv, bn, a, geom, relu, residual=None, kind=None):
        w = conv.weight.data
        cout, cin, kh, _ = w.shape
        pi = None
        if self._plan_live is not None:
            plan, idx = self._plan_live
            pi = idx.get(id(conv))
        if pi is not None:
            gamma, beta, scale, shift = self.master_of(bn.weight), self.master_of(bn.bias), plan.scale[pi], plan.shift[pi]
        else:
            gamma, beta, scale, shift = self._bn_vectors(bn)
        if kind == "stem":
            w2 = torch.zeros(cout, 64, dtype=BF16, device=w.device)
            w2[:, :27] = w.reshape(cout, 27)
            wop, convarg = RawWeight(w2, bias=shift, K=32), None
        elif kh == 1:
            wop, convarg = RawWeight(plan.fwd[pi] if pi is not None else ops.conv_weight_relayout(w, 0), bias=shift, K=cin), None
        else:
            wop = RawWeight(plan.fwd[pi] if pi is not None else ops.conv_weight_relayout(w, 0), bias=shift, K=9 * cin)
            convarg = (geom[1], geom[2], cin)
        rec = {"conv": conv, "bn": bn, "a": a, "geom": geom, "kind": kind or ("1x1" if kh == 1 else "3x3"),
               "gamma": gamma, "beta": beta, "sub": residual, "dgrad_op": plan.dgrad[pi] if pi is not None else None}
        if self.bn_batch_stats:
            # batch statistics: raw conv output, per-channel sums, fold, normalise (+ residual, ReLU); the running
            # statistics (fp32 copies) are updated in place with nn.BatchNorm2d's momentum rule
            z = ops.gemm(a, wop, conv=convarg, layout="rm", use_bias=False)
            M, C = z.shape
            sums = torch.zeros(2, C, dtype=F32, device=z.device)
            ops.colsum(z, sums[0])
            ops.colsum(z, sums[1], z)
            st = self._bn_stats[id(bn)]
            mom = 0.1 if bn.momentum is None else float(bn.momentum)
            scale, shift, mean, rstd = ops.bn_batch_fold(sums[0], sums[1], gamma, beta, M, bn.eps, mom, st[0], st[1])
            self._bn_dirty = True
            y = ops.bn_apply(z, scale, shift, res=residual, relu=relu or residual is not None)
            rec.update(y=y, z=z, mean=mean, rstd=rstd, scale=None, batch=True)
            return y, rec
        if residual is None:
            y = ops.gemm(a, wop, scale=scale, act=ops.MG_ACT_RELU if relu else ops.MG_ACT_NONE, conv=convarg, layout="rm")
        else:
            y = ops.gemm(a, wop, scale=scale, residuals=(residual,), act_after=ops.MG_ACT_RELU, conv=convarg, layout="rm")
        rec.update(y=y, scale=scale, batch=False)
        return y, rec

    @staticmethod
    def _pad_k(w2):
        n, k = w2.shape
        kp = ops.ceil_to(k, 64)
        if kp == k:
            return w2.contiguous()
        out = torch.zeros(n, kp, dtype=w2.dtype, device=w2.device)
        out[:, :k] = w2
        return out

    def _encoder_forward(self, images):
        enc = self.module.image_prefix.enc
        if not any(self.is_trainable(p) for p in enc.parameters()):
            return enc(images), None           # frozen trunk: inference path, nothing taped
        x = images.to(BF16).contiguous()
        B, _, H, W = x.shape
        h, w = H // 2, W // 2
        units = []
        self._plan_live = None
        if _CONV_PLAN and not self.bn_batch_stats:      # batch statistics: the dgrad operand is unscaled and the affine comes from the batch
            self._plan_live = self._conv_plan(enc)
            self._plan_live[0].refresh()
        cols = ops.stem_im2col(x)
        y, r = self._unit_fwd(enc.conv1, enc.bn1, cols, (B, h, w), True, kind="stem"); units.append(r)
        y, r = self._unit_fwd(enc.conv2, enc.bn2, y, (B, h, w), True); units.append(r)
        y, r = self._unit_fwd(enc.conv3, enc.bn3, y, (B, h, w), True); units.append(r)
        stem_out = y
        y = ops.avgpool2(y.view(B, h, w, -1))
        h, w = h // 2, w // 2
        y = y.view(B * h * w, -1)
        blocks = []
        for li in range(1, 5):
            for blk in getattr(enc, f"layer{li}"):
                br = {"x": y, "geom": (B, h, w), "stride": blk.stride}
                o1, br["u1"] = self._unit_fwd(blk.conv1, blk.bn1, y, (B, h, w), True)
                o2, br["u2"] = self._unit_fwd(blk.conv2, blk.bn2, o1, (B, h, w), True)
                p, xi = o2, y
                if blk.stride > 1:
                    p = ops.avgpool2(o2.view(B, h, w, -1)).view(B * (h // 2) * (w // 2), -1)
                    xi = ops.avgpool2(y.view(B, h, w, -1)).view(B * (h // 2) * (w // 2), -1)
                    h, w = h // 2, w // 2
                identity = y if blk.downsample is None else None
                if blk.downsample is not None:
                    identity, br["ud"] = self._unit_fwd(blk.downsample[1], blk.downsample[2], xi, (B, h, w), False)
                y, br["u3"] = self._unit_fwd(blk.conv3, blk.bn3, p, (B, h, w), False, residual=identity)
                br["out_geom"] = (B, h, w)
                blocks.append(br)
        feats = y.view(B, h * w, -1)
        return feats, {"units": units, "stem_out": stem_out, "blocks": blocks, "B": B}

    def _unit_bwd(self, rec, g, need_dgrad=True, gate=None, residuals=(), gate_after=False):
        """g: gradient wrt the unit's BN output, already ReLU-gated, [M, Cout].
        Accumulates conv / BN-affine gradients; returns the gradient wrt the unit's
        input, gated by ``gate`` (> 0) and with ``residuals`` added in the epilogue."""
        conv, bn = rec["conv"], rec["bn"]
        w = conv.weight.data
        cout, cin, kh, _ = w.shape
        Bq, hh, ww = rec["geom"]
        a, scale = rec["a"], rec["scale"]
        if rec["batch"]:
            # dbeta = sum g, dgamma = sum g * xhat = rstd * (sum g*z - mean * sum g); then the gradient wrt the raw conv output
            C = g.shape[1]
            sums = torch.zeros(2, C, dtype=F32, device=g.device)
            ops.colsum(g, sums[0])
            ops.colsum(g, sums[1], rec["z"])
            dbeta = sums[0]
            dgamma = rec["rstd"] * (sums[1] - rec["mean"] * dbeta)
            self.grad_of(bn.weight).add_(dgamma)
            self.grad_of(bn.bias).add_(dbeta)
            g = ops.bn_bwd_dz(g, rec["z"], rec["mean"], rec["rstd"], rec["gamma"], dgamma, dbeta)
        # the unit's parameter gradients: BatchNorm affine (frozen statistics) + convolution weight
        if rec["batch"]:
            gT = _t(g)
        elif _BN_GRAD_FUSED and g.is_contiguous() and rec["y"].is_contiguous() and (rec["sub"] is None or rec["sub"].is_contiguous()):
            # g^T (the weight gradient's operand) and the BatchNorm parameter gradients from ONE pass over g
            gT = RawWeight(ops.transpose_bn_param_grad(g, rec["y"], rec["sub"], rec["gamma"], rec["beta"], self.grad_of(bn.weight), self.grad_of(bn.bias)))
        else:
            ops.bn_param_grad(g, rec["y"], rec["sub"], rec["gamma"], rec["beta"], self.grad_of(bn.weight), self.grad_of(bn.bias))
            gT = _t(g)
        if rec["kind"] == "3x3":
            # im2col^T rows come in the weight's own (ci, ky, kx) order -> dW needs no re-layout
            self._acc_wgrad(conv.weight, gT, RawWeight(ops.im2col_t(a, Bq, hh, ww, cin)), row_scale=scale)
        else:   # 1x1, or the stem's explicit im2col matrix (columns already in (c, ky, kx) order)
            self._acc_wgrad(conv.weight, gT, _t(a), row_scale=scale)
        if not need_dgrad:
            return None
        aux_kw = {}
        if gate is not None:
            aux_kw = dict(aux=gate, aux_mode=ops.MG_AUX_RELU_GATE, aux_after=gate_after)
        dg = rec.get("dgrad_op")       # made at the start of the forward by the batched re-layout (frozen statistics)
        if dg is None:
            dg = ops.conv_weight_relayout(w, 1, scale)
        if rec["kind"] == "3x3":
            # dX = conv3x3(g, W') with W'[ci][(ky,kx),co] = W[co][ci][2-ky][2-kx] * scale[co]
            wop = RawWeight(dg, K=9 * cout)
            return ops.gemm(g, wop, conv=(hh, ww, cout), layout="rm", use_bias=False, residuals=residuals, **aux_kw)
        wop = RawWeight(dg, K=cout)     # [cin, cout] * scale[co]
        return ops.gemm(g, wop, layout="rm", use_bias=False, residuals=residuals, **aux_kw)

    def _encoder_backward(self, et, g_out):
        """g_out: gradient wrt the trunk output, already gated by (out > 0).  Each stage's parameters (layer4 first, the
        stem last) are handed to the gradient exchange as soon as its first block has been back-propagated, so the trunk's
        136 M-element bucket travels under the rest of the trunk's backward instead of in step()."""
        blocks = et["blocks"]
        enc = self.module.image_prefix.enc
        stage_start, n = {}, 0
        for li in range(1, 5):
            stage_start[n] = li
            n += len(getattr(enc, f"layer{li}"))
        g3 = g_out
        for bi in range(len(blocks) - 1, -1, -1):
            br = blocks[bi]
            B, h, w = br["geom"]
            x = br["x"]
            prev_is_relu = bi > 0            # block input is the previous block's ReLU output (not for layer1.0)
            # identity path
            if "ud" in br:
                dxi = self._unit_bwd(br["ud"], g3)
                d_id = ops.avgpool2_bwd(dxi.view(B, h // 2, w // 2, -1), B, h, w, dxi.shape[1]).view(B * h * w, -1) \
                    if br["stride"] > 1 else dxi
            else:
                d_id = g3
            # main path: conv3 -> (pool) -> conv2 -> conv1
            u2y = br["u2"]["y"]
            if br["stride"] > 1:
                dp = self._unit_bwd(br["u3"], g3)
                g2 = ops.avgpool2_bwd(dp.view(B, h // 2, w // 2, -1), B, h, w, dp.shape[1], gate=u2y).view(B * h * w, -1)
            else:
                g2 = self._unit_bwd(br["u3"], g3, gate=u2y)
            g1 = self._unit_bwd(br["u2"], g2, gate=br["u1"]["y"])
            g3 = self._unit_bwd(br["u1"], g1, gate=x if prev_is_relu else None, residuals=(d_id,), gate_after=True)
            blocks[bi] = None
            if bi in stage_start:
                self._reduce_params_async([p for p in getattr(enc, f"layer{stage_start[bi]}").parameters() if self.is_trainable(p)])
        # stem: avgpool -> conv3 -> conv2 -> conv1
        units = et["units"]
        B, h, w = units[2]["geom"]
        g = ops.avgpool2_bwd(g3.view(B, h // 2, w // 2, -1), B, h, w, g3.shape[1], gate=et["stem_out"]).view(B * h * w, -1)
        g = self._unit_bwd(units[2], g, gate=units[1]["y"])
        g = self._unit_bwd(units[1], g, gate=units[0]["y"])
        self._unit_bwd(units[0], g, need_dgrad=False)
        self._reduce_params_async([p for m in (enc.conv1, enc.bn1, enc.conv2, enc.bn2, enc.conv3, enc.bn3)
                                   for p in m.parameters() if self.is_trainable(p)])

    # ---- optimizer step ----------------------------------------------------------------
    def _step_impl(self):
        if self.micro_steps % self.gas != 0:
            return
        self.global_steps += 1
        lrs = self.lr_scheduler.get_lr()
        grad_scale = 1.0 / self.gas
        if self._dist:
            # gradient average over ranks (DeepSpeed ZeRO-2 reduce-scatter semantics: mean), RCCL over xGMI;
            # most buckets were launched from backward() and have been running under it
            if self.time_comm:       # exposed exchange time: what of the all-reduce backward did NOT cover (bench.py reads it)
                ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                ev[0].record()
            self._finish_reduce()
            if self.time_comm:
                ev[1].record()
                self._comm_events.append(ev)
                self._busy_steps += 1
            grad_scale /= self.world
        self._norm_sq.zero_()
        grads = [g.comm if self.exchange_bf16 else g.grad for g in self.groups]    # what came back from the exchange
        for gr in grads:
            ops.sumsq(gr, self._norm_sq)
        for g, gr, lr in zip(self.groups, grads, lrs):
            ops.adamw(g.master, g.m, g.v, gr, g.model, lr, self.betas[0], self.betas[1], self.eps, g.wd,
                      self.global_steps, max_norm=self.clip, norm_sq=self._norm_sq, grad_scale=grad_scale)
            g.grad.zero_()
        self.lr_scheduler.step()
        from .adapters import bump_weights_epoch
        bump_weights_epoch()        # the AdamW above wrote the parameters through raw pointers: Adapter.forward's packs are stale
        self.module.image_prefix.invalidate_packed()
        self._adapters_dirty = True
        if self.lm_trainable:       # the forward / dgrad operands are packed COPIES of the LM weights: rebuild them from the new values
            self.module.lm.invalidate_packed()
            self._lm_train_packs = None
            self._adapters_dirty = False

    def grad_norm(self) -> float:
        return float(torch.sqrt(self._norm_sq)) / self.gas / self.world

    # ---- DeepSpeed-protocol odds and ends -------------------------------------------------
    def deepspeed_io(self, dataset, batch_size=None, collate_fn=None, num_workers=None, pin_memory=True):
        """The loader DeepSpeed's engine hands out (reference train.py:103-112).  For a dataset that reads image files, worker
        processes decode / resize on the host while the step runs (MAGMA_LOADER_WORKERS, default min(8, host cores / ranks);
        they iterate the dataset's host-side view -- same pixels, no GPU in a forked worker), batches arrive in pinned memory
        (the engine's non-blocking copies overlap the step) from one persistent worker pool.  Without workers (synthetic data,
        MAGMA_LOADER_WORKERS=0) the items are produced on the launch thread, RGB images through the device-side resampler.
        DistributedSampler shards per rank."""
        from torch.utils.data import DataLoader
        from torch.utils.data.distributed import DistributedSampler
        bs = batch_size or max(1, self.config.batch_size // (self.gas * self.world))
        sampler = DistributedSampler(dataset) if self.world > 1 else None
        from .datasets import collate_fn as default_collate, host_side_view, on_disk
        from functools import partial
        if num_workers is None:
            env = os.environ.get("MAGMA_LOADER_WORKERS")
            num_workers = int(env) if env is not None else (min(8, max(1, (os.cpu_count() or 2) // max(1, self.world)))
                                                           if on_disk(dataset) else 0)
        kw = {}
        if num_workers > 0:
            dataset = host_side_view(dataset)
            # workers are SPAWNED, not forked: a fork of the training process carries the HIP runtime's and RCCL's state into a child
            # that must not touch the GPU, and forking a process with tens of GB mapped took 20 x longer on some hosts than on others
            # (the same loader test: 3 s and 70 s).  The dataset view, its transform and the collate function travel by pickle.
            kw = dict(num_workers=num_workers, persistent_workers=True, prefetch_factor=2, multiprocessing_context="spawn",
                      pin_memory=bool(pin_memory) and torch.cuda.is_available())
        return DataLoader(dataset, batch_size=bs, sampler=sampler, shuffle=False,
                          collate_fn=collate_fn or partial(default_collate, seq_len=self.module.seq_len), **kw)

    def _param_names(self) -> Dict[int, str]:
        return {id(p): n for n, p in self.module.named_parameters()}

    def save_checkpoint(self, save_dir, client_state=None, tag=None):
        """DeepSpeed layout: <dir>/<tag>/mp_rank_00_model_states.pt with {"module": state_dict, ...} + <dir>/latest."""
        tag = tag or f"global_step{self.global_steps}"
        self._flush_bn_stats()
        if (not dist.is_initialized()) or dist.get_rank() == 0:
            path = Path(save_dir) / tag
            path.mkdir(parents=True, exist_ok=True)
            sd = {"module": {k: v.detach().cpu() for k, v in self.module.state_dict().items()},
                  "optimizer": {"format": "per_parameter_v1",
                                "state": {k: v for g in self.groups for k, v in g.state_by_name(self._param_names()).items()}},
                  "lr_scheduler": self.lr_scheduler.state_dict(), "global_steps": self.global_steps,
                  "micro_steps": self.micro_steps}
            sd.update(client_state or {})
            torch.save(sd, path / "mp_rank_00_model_states.pt")
            (Path(save_dir) / "latest").write_text(tag)
        if dist.is_initialized():
            dist.barrier()

    def load_checkpoint(self, load_dir, load_optimizer_states=True, load_lr_scheduler_states=True, tag=None):
        latest = Path(load_dir) / "latest"
        if tag is None:
            if not latest.exists():
                return None, None
            tag = latest.read_text().strip()
        path = Path(load_dir) / tag / "mp_rank_00_model_states.pt"
        if not path.exists():
            return None, None
        sd = torch.load(path, map_location="cpu", weights_only=False)
        self.module.load_checkpoint_state(sd["module"])
        self._bn_stats = {}         # BatchNorm statistics may have changed
        self._plan = self._plan_live = None
        for g in self.groups:       # parameters were re-pointed at the flat buffers; refresh masters from the loaded values
            for p, o in zip(g.params, g.offsets):
                g.master[o:o + p.numel()].copy_(p.data.reshape(-1).float())
        if load_optimizer_states and "optimizer" in sd:
            opt = sd["optimizer"]
            if isinstance(opt, dict) and opt.get("format") == "per_parameter_v1":
                names = self._param_names()
                missing = [n for g in self.groups for n in g.load_state_by_name(opt["state"], names)]
                if missing:
                    raise RuntimeError(f"optimizer state of the checkpoint lacks / mismatches {len(missing)} trainable tensors "
                                       f"(first: {missing[:3]}); pass load_optimizer_states=False to restart the optimizer")
            else:      # rounds 1-2 wrote the padded flat buffers verbatim: usable only if the layout still matches
                for g, s in zip(self.groups, opt):
                    if s["master"].numel() != g.master.numel():
                        raise RuntimeError("optimizer state in the legacy flat layout does not match this build's buffer "
                                           "layout; pass load_optimizer_states=False to restart the optimizer")
                    g.master.copy_(s["master"]); g.m.copy_(s["m"]); g.v.copy_(s["v"])
            for g in self.groups:
                g.model.copy_(g.master)
            self.global_steps = sd.get("global_steps", 0)
            self.micro_steps = sd.get("micro_steps", 0)
        if load_lr_scheduler_states and "lr_scheduler" in sd:
            self.lr_scheduler.load_state_dict(sd["lr_scheduler"])
        self.module.invalidate_packed()
        self._lm_train_packs = None        # transposed dgrad copies / e4m3 copies of the old frozen weights are stale now
        self._fp8_packs = {}
        self._out_up = {}
        return str(path), sd


def initialize(model, config=None, model_parameters=None, training_data=None, collate_fn=None, **unused):
    """deepspeed.initialize(...) look-alike: returns (engine, optimizer, train_loader, lr_scheduler)."""
    engine = MagmaEngine(model, config, param_groups=model_parameters)
    loader = engine.deepspeed_io(training_data, collate_fn=collate_fn) if training_data is not None else None
    return engine, engine, loader, engine.lr_scheduler
