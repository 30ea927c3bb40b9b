"""mcx_dep_flags_noalloc_dev (the entry point that allocates nothing: its LOADED
bitmap is a static device array) against mcx_dep_flags_dev, through the C ABI,
bit for bit, on two calls in a row of which the second has the smaller graph."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(ctx, synth, seed, V, deg):
    rowptr, col = synth.dep_graph(seed, V, deg)
    flags = synth.dep_flags(seed, rowptr, col)
    ctx.dep_graph_load(rowptr, col)
    d_in = ctx.to_device(flags)
    outs = []
    for fn in (ctx.dep_flags_noalloc_dev, ctx.dep_flags_dev):
        d_out, d_ref = ctx.alloc(V), ctx.alloc(4 * V)
        fn(d_in.ptr, d_out.ptr, d_ref.ptr)
        ctx.sync()
        outs.append((d_out.download(np.uint8, V), d_ref.download(np.uint32, V)))
        d_out.free()
        d_ref.free()
    d_in.free()
    return outs, (rowptr, col, flags)


def test_noalloc_matches_dep_flags_twice(gpu_ctx, oracle_mod):
    from montecarlo_amd import synth

    # first a graph of 200 003 modules (the bitmap ends in the middle of a word and
    # of a 16-byte piece), then a much smaller one: what the first call left in
    # the static array must not show in the second
    for seed, V, deg in ((11, 200_003, 8), (12, 2_501, 5)):
        (got, want), (rowptr, col, flags) = _both(gpu_ctx, synth, seed, V, deg)
        assert np.array_equal(got[0], want[0]), f"flags differ at V={V}"
        assert np.array_equal(got[1], want[1]), f"refcnt differs at V={V}"
        o_out, o_ref = oracle_mod.dep_flags(rowptr, col, flags)
        assert np.array_equal(got[0], o_out) and np.array_equal(got[1], o_ref)
        assert int((got[0] & 1).sum()) > 0  # the case is not trivially empty


def test_noalloc_without_refcnt_and_unaligned_flags(gpu_ctx):
    from montecarlo_amd import synth

    V = 70_001
    rowptr, col = synth.dep_graph(13, V, 6)
    flags = synth.dep_flags(13, rowptr, col)
    gpu_ctx.dep_graph_load(rowptr, col)
    d_raw = gpu_ctx.alloc(V + 16)
    padded = np.zeros(V + 16, dtype=np.uint8)
    padded[3:3 + V] = flags
    d_raw.upload(padded)
    d_a, d_b = gpu_ctx.alloc(V), gpu_ctx.alloc(V)
    gpu_ctx.dep_flags_noalloc_dev(d_raw.ptr + 3, d_a.ptr, None)
    gpu_ctx.dep_flags_dev(d_raw.ptr + 3, d_b.ptr, None)
    gpu_ctx.sync()
    assert np.array_equal(d_a.download(np.uint8, V), d_b.download(np.uint8, V))
    for b in (d_raw, d_a, d_b):
        b.free()


def test_noalloc_refuses_graphs_beyond_the_static_bitmap(gpu_ctx):
    from montecarlo_amd import binding

    V = (4 << 20) + 1
    rowptr = np.zeros(V + 1, dtype=np.uint32)
    col = np.zeros(0, dtype=np.uint32)
    gpu_ctx.dep_graph_load(rowptr, col)
    d_in, d_out = gpu_ctx.alloc(V), gpu_ctx.alloc(V)
    with pytest.raises(binding.McxError) as e:
        gpu_ctx.dep_flags_noalloc_dev(d_in.ptr, d_out.ptr, None)
    assert e.value.rc == -6
    d_in.free()
    d_out.free()


def _hip():
    """the HIP runtime the library itself is linked to, for the two things the test needs to
    see from outside: free device memory and stream capture"""
    import ctypes

    hip = ctypes.CDLL("libamdhip64.so")
    vp, sz = ctypes.c_void_p, ctypes.c_size_t
    hip.hipMemGetInfo.argtypes = [ctypes.POINTER(sz), ctypes.POINTER(sz)]
    hip.hipStreamBeginCapture.argtypes = [vp, ctypes.c_int]
    hip.hipStreamEndCapture.argtypes = [vp, ctypes.POINTER(vp)]
    hip.hipGraphInstantiate.argtypes = [ctypes.POINTER(vp), vp, vp, vp, sz]
    hip.hipGraphLaunch.argtypes = [vp, vp]
    hip.hipGraphExecDestroy.argtypes = [vp]
    hip.hipGraphDestroy.argtypes = [vp]
    return hip


def test_noalloc_allocates_nothing_and_can_be_captured(gpu_ctx):
    """What the entry point is for: (1) free device memory is the same before and after a run
    of calls; (2) the call is legal between hipStreamBeginCapture and hipStreamEndCapture on
    the context's stream - where an allocation, a free or a synchronisation would end the
    capture with an error - and the captured graph, launched, gives mcx_dep_flags_dev's
    results."""
    import ctypes

    from montecarlo_amd import synth

    hip = _hip()
    V = 120_001
    rowptr, col = synth.dep_graph(14, V, 7)
    flags = synth.dep_flags(14, rowptr, col)
    gpu_ctx.dep_graph_load(rowptr, col)
    d_in = gpu_ctx.to_device(flags)
    d_out, d_ref, d_want, d_wref = gpu_ctx.alloc(V), gpu_ctx.alloc(4 * V), gpu_ctx.alloc(V), gpu_ctx.alloc(4 * V)
    gpu_ctx.dep_flags_dev(d_in.ptr, d_want.ptr, d_wref.ptr)
    gpu_ctx.dep_flags_noalloc_dev(d_in.ptr, d_out.ptr, d_ref.ptr)   # first call: whatever is set up lazily
    gpu_ctx.sync()
    free0, free1, total = ctypes.c_size_t(), ctypes.c_size_t(), ctypes.c_size_t()
    assert hip.hipMemGetInfo(ctypes.byref(free0), ctypes.byref(total)) == 0
    for _ in range(50):
        gpu_ctx.dep_flags_noalloc_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
    gpu_ctx.sync()
    assert hip.hipMemGetInfo(ctypes.byref(free1), ctypes.byref(total)) == 0
    assert free1.value == free0.value, "the call changed the amount of free device memory"
    # under capture
    stream = gpu_ctx.lib.mcx_ctx_stream(gpu_ctx.h)
    zeros = np.zeros(4 * V, dtype=np.uint8)
    d_out.upload(zeros[:V])
    d_ref.upload(zeros)
    gpu_ctx.sync()
    graph, graph_exec = ctypes.c_void_p(), ctypes.c_void_p()
    assert hip.hipStreamBeginCapture(stream, 0) == 0          # hipStreamCaptureModeGlobal
    try:
        gpu_ctx.dep_flags_noalloc_dev(d_in.ptr, d_out.ptr, d_ref.ptr)
    finally:
        rc_end = hip.hipStreamEndCapture(stream, ctypes.byref(graph))
    assert rc_end == 0 and graph.value, f"capture ended with HIP error {rc_end}"
    # nothing ran yet: the outputs are still zero
    gpu_ctx.sync()
    assert not d_out.download(np.uint8, V).any()
    assert hip.hipGraphInstantiate(ctypes.byref(graph_exec), graph, None, None, 0) == 0
    for _ in range(2):
        assert hip.hipGraphLaunch(graph_exec, stream) == 0
    gpu_ctx.sync()
    assert np.array_equal(d_out.download(np.uint8, V), d_want.download(np.uint8, V))
    assert np.array_equal(d_ref.download(np.uint32, V), d_wref.download(np.uint32, V))
    assert hip.hipGraphExecDestroy(graph_exec) == 0 and hip.hipGraphDestroy(graph) == 0
    for b in (d_in, d_out, d_ref, d_want, d_wref):
        b.free()
