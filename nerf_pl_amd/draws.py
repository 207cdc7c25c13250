"""The random draws of a training step in ONE launch, from torch's own generator stream (csrc/draws.hip).

The reference draws inside `render_rays` with `torch.rand` / `torch.randn` (rendering.py:203, :152, :39, :152) and picks its batch
through the DataLoader (train.py:89-94); on the GPU that is five ~5 us launches per step.  `draws()` produces the same tensors —
bit for bit what those torch calls return for the generator's current (seed, offset) — from one HIP launch and advances the
generator by what the calls would have consumed, so a run is the same run whichever path draws (`tests/test_gpu_draws.py`).

Self-check.  The launch REPLICATES ATen's Philox consumption (offset arithmetic, Box-Muller, the 32- / 64-bit randint paths) —
pinned by the GPU tests on torch 2.10 + ROCm 7.0, but a torch upgrade could move it silently.  The first use on a device
therefore draws the same short uniform / normal / integer sequences through torch and through the replica from two private
generators at the same (seed, offset) and compares values and offsets bit for bit (`replica_ok`); on a mismatch this module
warns once and makes every draw with torch's own calls from then on (same values as an all-torch run, the reference's launch
count: the batch's rays and the weight images then come from their stand-alone launches).  NERFHIP_DRAWS=torch forces that path,
NERFHIP_DRAWS=replica skips the check.

hipGraph capture.  A captured launch cannot take (seed, offset) by value: every replay must walk on.  While the current stream
is capturing, `draws()` reads them from a device-resident `GraphDrawState` that the kernel itself advances (last workgroup,
arrival ticket); the owner of the graph (`system.GraphedTrainStep`) arms its state before the capture, captures under
`with capturing(state)` and moves torch's generator along after every replay (`after_replay`), so eager draws made between
replays stay on the same stream.
"""
import ctypes
import os
import warnings

import torch

from . import _lib
from ._lib import DRAW_NORMAL, DRAW_RANDINT, DRAW_UNIFORM, check, ptr, stream_ptr

_KINDS = {"rand": DRAW_UNIFORM, "randn": DRAW_NORMAL, "randint": DRAW_RANDINT}
_MAX_BLOCKS = {}
_GRAPH_STATES = {}
_REPLICA = {}          # device index -> True: the replica reproduces this torch build's streams | False: torch's own draws
# the self-check's draws: shapes that take the single- and the multi-block path, both randint widths (high < / >= 2^28)
_CHECK_SPECS = (("randint", (1000,), 64000000), ("rand", (257, 64)), ("randn", (257, 192)), ("randint", (129,), 1 << 40),
                ("rand", (3,)), ("randn", (5, 7)))


def max_blocks(device):
    """ATen's launch cap for its distribution kernels: multiProcessorCount * (maxThreadsPerMultiProcessor / 256)."""
    idx = torch.device(device).index
    if idx is None:
        idx = torch.cuda.current_device()
    mb = _MAX_BLOCKS.get(idx)
    if mb is None:
        p = torch.cuda.get_device_properties(idx)
        mb = _MAX_BLOCKS[idx] = p.multi_processor_count * (p.max_threads_per_multi_processor // 256)
    return mb


def _generator(device, generator):
    if generator is not None:
        return generator
    torch.cuda.init()                      # (the tuple of default generators is empty until the runtime is initialised)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    return torch.cuda.default_generators[idx]


def _index(device):
    device = torch.device(device)
    return device.index if device.index is not None else torch.cuda.current_device()


def _torch_draw(spec, device, generator):
    kind, shape = spec[0], tuple(int(x) for x in spec[1])
    if kind == "randint":
        return torch.randint(0, int(spec[2]), shape, device=device, generator=generator)
    return (torch.rand if kind == "rand" else torch.randn)(*shape, device=device, generator=generator)


def _self_check(device):
    """torch's own rand / randn / randint against the replica, from two private generators at the same (seed, offset)."""
    device = torch.device("cuda", _index(device))
    seed, off = 0x5eed5eed, 4 * 977
    g_t, g_r = torch.Generator(device=device), torch.Generator(device=device)
    for g in (g_t, g_r):
        g.manual_seed(seed)
        g.set_offset(off)
    want = [_torch_draw(sp, device, g_t) for sp in _CHECK_SPECS]
    got = _replica_draws(list(_CHECK_SPECS), device, g_r, None, None)
    same = all(torch.equal(a, b) for a, b in zip(want, got))
    return bool(same) and int(g_t.get_offset()) == int(g_r.get_offset())


def replica_ok(device):
    """Does the Philox replica reproduce THIS torch build's generator streams on `device`?  Checked once per device, outside
    any capture (GraphDrawState.arm() asks before a capture starts)."""
    idx = _index(device)
    ok = _REPLICA.get(idx)
    if ok is None:
        mode = os.environ.get("NERFHIP_DRAWS", "check")
        if mode == "torch":
            ok = False
        elif mode == "replica":
            ok = True
        else:
            if torch.cuda.is_current_stream_capturing():
                raise _lib.NerfHipError("draws: first use on this device inside a hipGraph capture — call draws.replica_ok(device) "
                                        "(or GraphDrawState.arm()) before capturing")
            ok = _self_check(device)
            if not ok:
                warnings.warn("nerf_pl_amd.draws: the Philox replica does not reproduce torch %s's rand / randn / randint stream "
                              "on this device; falling back to torch's own draws (separate launches, same values as an "
                              "all-torch run)" % torch.__version__)
        _REPLICA[idx] = ok
    return ok


class GraphDrawState:
    """(seed, offset) of one generator in device memory, for captured launches: {seed, offset, arrival ticket, -} as 4 x int64."""

    def __init__(self, device, generator=None):
        self.device = torch.device(device)
        self.generator = _generator(self.device, generator)
        self.tensor = torch.zeros(4, device=self.device, dtype=torch.int64)
        self.seed = None
        self.expect = None         # offset the device copy holds (== the generator's, as long as nobody else drew)
        self.increment = 0         # what ONE replay of the captured launches consumes

    def arm(self):
        """Before a capture (not during one): load the generator's current (seed, offset); forget the recorded increment."""
        g = self.generator
        self.seed, self.expect = int(g.initial_seed()), int(g.get_offset())
        self.increment = 0
        self._upload()
        replica_ok(self.device)            # (the self-check cannot run once the capture has begun)

    def _upload(self):
        wrap = lambda v: v - (1 << 64) if v >= (1 << 63) else v          # uint64 bit patterns in an int64 tensor
        host = torch.tensor([wrap(self.seed), wrap(self.expect), 0, 0], dtype=torch.int64)
        self.tensor.copy_(host)

    def before_replay(self):
        """The generator moved since the last replay (an eager draw, a re-seed): put the device copy back on its stream."""
        g = self.generator
        seed, off = int(g.initial_seed()), int(g.get_offset())
        if seed != self.seed or off != self.expect:
            self.seed, self.expect = seed, off
            self._upload()

    def after_replay(self):
        """The replay advanced the device copy by `increment`: torch's generator follows (a host-side counter, no launch)."""
        if self.increment:
            self.expect += self.increment
            self.generator.set_offset(self.expect)


class capturing:
    """`with capturing(state):` — the draw launches issued (captured) inside read their generator state from `state` and record
    their increments there.  One state per captured graph: the graph's launches hold ITS device buffer."""

    def __init__(self, state):
        self.state = state

    def __enter__(self):
        key = self.state.device.index if self.state.device.index is not None else torch.cuda.current_device()
        self.key, self.prev = key, _GRAPH_STATES.get(key)
        _GRAPH_STATES[key] = self.state
        return self.state

    def __exit__(self, *exc):
        if self.prev is None:
            _GRAPH_STATES.pop(self.key, None)
        else:
            _GRAPH_STATES[self.key] = self.prev
        return False


def _capture_state(device):
    return _GRAPH_STATES.get(device.index if device.index is not None else torch.cuda.current_device())


def increment(numel, device):
    """What one torch.rand / randn / randint call of `numel` elements advances a GPU generator's offset by."""
    return int(_lib.load().nerfhip_torch_draw_increment(int(numel), max_blocks(device)))


def draws(specs, device, generator=None, batch=None, pack=None):
    """specs: [("rand", shape) | ("randn", shape) | ("randint", shape, high)], at most 6, each optionally followed by False =
    "nobody reads this draw" — returns the tensors the torch calls `torch.rand(*shape, device=device)`, ... would return IN THIS
    ORDER for `generator` (default: the device's default generator; None for the unread ones), which is advanced exactly as by
    those calls.  batch: an _lib.RayBatch for specs[0] (a randint over a RayStore's pixel ids): the drawn ids become rays / rgbs
    in the same launch (rays.RayStore.sample) and are themselves not stored when specs[0] is marked unread.
    pack = (models, mlp_dtype): the launch also packs those models' weight images (nerfhip_train_prologue: the whole prologue of
    a training step in one launch); the buffers are the models' own (ops.pack_models_train's)."""
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.NerfHipError("nerf_pl_amd.draws runs on MI355X only (no CPU fallback); use torch.rand / randn on CPU tensors")
    n = len(specs)
    if not 1 <= n <= 6:
        raise ValueError("draws: 1..6 draws per launch")
    if not replica_ok(device):
        return _torch_draws(specs, device, generator, batch, pack)
    return _replica_draws(specs, device, generator, batch, pack)


def _torch_draws(specs, device, generator, batch, pack):
    """The same tensors from torch's own calls, in order (unread draws are made and dropped: the stream must advance as the
    reference's does); the batch's rays / colours and the weight images from their stand-alone launches."""
    from . import ops
    outs = []
    with torch.cuda.device(device):
        for sp in specs:
            kind = _KINDS[sp[0]]
            live = not (len(sp) > (3 if kind == DRAW_RANDINT else 2) and sp[-1] is False)
            t = _torch_draw(sp, device, generator)
            outs.append(t if live or (batch is not None and not outs) else None)
        if batch is not None:
            ids = outs[0].reshape(-1).contiguous()
            check(_lib.load().nerfhip_sample_batch(batch.c2w, ptr(ids), batch.rgbs_all, ids.numel(), batch.H, batch.W, batch.focal,
                                                   batch.near, batch.far, batch.use_ndc, batch.ndc_near_plane, batch.rays, batch.rgbs,
                                                   stream_ptr()), "nerfhip_sample_batch")
            sp = specs[0]
            if len(sp) > 3 and sp[-1] is False:
                outs[0] = None
    if pack is not None:
        ops.pack_models_train(*pack)
    return outs


def _replica_draws(specs, device, generator, batch, pack):
    n = len(specs)
    outs, arr = [], (_lib.Draw * n)()
    with torch.cuda.device(device):
        for i, sp in enumerate(specs):
            kind = _KINDS[sp[0]]
            shape = tuple(int(x) for x in sp[1])
            numel = 1
            for x in shape:
                numel *= x
            live = not (len(sp) > (3 if kind == DRAW_RANDINT else 2) and sp[-1] is False)
            if kind == DRAW_RANDINT:
                high = int(sp[2])
                if not 1 <= high <= (1 << 62):
                    raise ValueError("draws: randint needs 1 <= high <= 2^62")
                arr[i].range = high
            t = torch.empty(shape, device=device, dtype=torch.int64 if kind == DRAW_RANDINT else torch.float32) if live else None
            outs.append(t)
            arr[i].kind, arr[i].numel, arr[i].out = kind, numel, (t.data_ptr() if t is not None else None)
        inc = ctypes.c_uint64(0)
        bptr = ctypes.addressof(batch) if batch is not None else None
        lib = _lib.load()
        if pack is not None:
            from . import ops
            models, dtype = pack
            W, Bv, P, Pb, _bufs = ops.pack_tables(models, dtype)
            tail = (W, Bv, P, Pb, len(models), ops.mlp_dtype_code(dtype), stream_ptr())

            def launch(seed, off, state):
                return lib.nerfhip_train_prologue(ctypes.addressof(arr), n, bptr, seed, off, state, max_blocks(device), ctypes.byref(inc),
                                                  *tail)
        else:
            def launch(seed, off, state):
                return lib.nerfhip_torch_draws(ctypes.addressof(arr), n, bptr, seed, off, state, max_blocks(device), ctypes.byref(inc),
                                               stream_ptr())
        if torch.cuda.is_current_stream_capturing():
            st = _capture_state(device)
            if st is None or st.seed is None or (generator is not None and generator is not st.generator):
                raise _lib.NerfHipError("draws() inside a hipGraph capture needs an armed GraphDrawState for its generator: capture "
                                        "under `with draws.capturing(state)` after `state.arm()` (system.GraphedTrainStep does both)")
            check(launch(0, 0, ptr(st.tensor)), "nerfhip_torch_draws")
            st.increment += int(inc.value)
        else:
            g = _generator(device, generator)
            seed, off = int(g.initial_seed()), int(g.get_offset())
            check(launch(seed, off, None), "nerfhip_torch_draws")
            g.set_offset(off + int(inc.value))
        if pack is not None:
            ops.mark_packed(pack[0])
    return outs


def in_graph_stream(device):
    """True while the current stream is capturing under an armed GraphDrawState for `device`: draws made now must come from
    draws() — the device-resident stream the captured batch / step draws walk — and not from torch's own capture-time generator
    bookkeeping, which starts every replay at the same offset as that state (the two would hand out the same Philox counters)."""
    return torch.cuda.is_current_stream_capturing() and _capture_state(torch.device(device)) is not None and replica_ok(device)


def rand(shape, device):
    """torch.rand(*shape, device=device) on the default generator's stream — through draws() inside a capture that owns a
    GraphDrawState (same values either way)."""
    if in_graph_stream(device):
        return draws([("rand", shape)], device)[0]
    return torch.rand(*shape, device=device)


def randn(shape, device):
    if in_graph_stream(device):
        return draws([("randn", shape)], device)[0]
    return torch.randn(*shape, device=device)


def step_specs(B, S, N, perturb, noise_std):
    """(keys, specs) of the four draws of one reference `render_rays` call in the reference's order, named as the
    parity tests name them: perturb_rand (B,S) if perturb > 0; noise_coarse (B,S) — ALWAYS drawn by the reference (rendering.py:152),
    read only when noise_std != 0; u (B,N) if N > 0 and perturb != 0; noise_fine (B,S+N) if N > 0, likewise."""
    keys, specs = [], []
    want_noise = noise_std != 0
    if perturb > 0:
        keys.append("perturb_rand"); specs.append(("rand", (B, S)))
    keys.append("noise_coarse"); specs.append(("randn", (B, S), want_noise))
    if N > 0:
        if perturb != 0:
            keys.append("u"); specs.append(("rand", (B, N)))
        keys.append("noise_fine"); specs.append(("randn", (B, S + N), want_noise))
    return keys, specs


def step_draws(B, S, N, perturb, noise_std, device, generator=None):
    """The draws of one `render_rays` call (step_specs) in ONE launch, as a dict; unread noise tensors are absent."""
    keys, specs = step_specs(B, S, N, perturb, noise_std)
    return {k: t for k, t in zip(keys, draws(specs, device, generator)) if t is not None}
