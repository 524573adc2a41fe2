"""hipGraph capture of the inference forward (eval, no_grad) — the launch-bound regime.

At <= 64 images per GPU the eval forward of MedMamba is bound by the host's launch rate (about 550 kernel launches,
~10 ms of host time against ~5 ms of GPU time at 32 images; tools/cpu_bound_check.py).  The forward has static shapes
and no host-side data dependence, so the whole launch sequence — the HIP kernels of this library (launched through the
C ABI on torch's current stream, which is the capture stream), the rocBLAS / hipBLASLt GEMMs and the MIOpen
convolutions, on both streams of the block schedule — is recorded once into a hipGraph and replayed with ONE launch.

    g = GraphedInference(net, example_batch)      # warms up, captures
    logits = g(batch)                             # copies into the static input, replays, returns the static output

Consumers this serves: test.py:76-108 / app_streamlit_demo.py:146-163 style evaluation loops over fixed-size batches.
Training steps are NOT captured: replay serialises the two-stream block schedule's kernels more than eager launching
does at 64 images per GPU.
"""
import torch


class GraphedInference:
    """Replays the eval forward of `net` from one hipGraph.

    The graph reads the model's raw parameters in place, but the folded BatchNorm / conv constants of every block
    (SS_Conv_SSM._eval_fold) are SEPARATE tensors that were built at capture time and are baked into the graph.  After
    `load_state_dict`, an optimizer step or any other in-place weight change a replay would therefore mix new raw weights with
    old folded ones.  `__call__` guards against that: it compares the version counters / storage addresses of all parameters
    and buffers with those seen at capture and recaptures (`recapture()`) when anything changed.  `check=False` skips the
    comparison (about 0.1 ms of host time per call) for callers that know the weights are frozen."""

    def __init__(self, net, example, warmup=3, check=True):
        if not example.is_cuda:
            raise RuntimeError("GraphedInference: HIP tensors only (there is no CPU path)")
        self.net = net.eval()
        self.static_in = example.detach().clone()
        self.warmup = max(1, warmup)
        self.check = check
        self.captures = 0
        self._capture()

    def _state_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.net.parameters()) + list(self.net.buffers()))

    def _capture(self):
        cur = torch.cuda.current_stream()
        side = torch.cuda.Stream(device=self.static_in.device)
        side.wait_stream(cur)
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(self.warmup):          # MIOpen solver selection, GEMM workspaces, BatchNorm folds: all before capture
                self.net(self.static_in)
        cur.wait_stream(side)
        torch.cuda.synchronize(self.static_in.device)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = self.net(self.static_in)
        self._key = self._state_key()
        self.captures += 1

    def recapture(self):
        """Drop every block's folded constants and record the graph again (after the weights changed)."""
        for m in self.net.modules():
            if hasattr(m, "_fold_cache"):
                m._fold_cache = None
        self.net.eval()
        self._capture()

    @torch.no_grad()
    def __call__(self, x):
        if x.shape != self.static_in.shape:
            raise RuntimeError(f"GraphedInference was captured for {tuple(self.static_in.shape)}, got {tuple(x.shape)}")
        if self.check and self._state_key() != self._key:
            self.recapture()
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out
