"""Introspection of a captured hipGraph through the HIP runtime (hipGraphGetNodes / hipGraphNodeGetType /
hipGraphDebugDotPrint): which node kinds a captured training step holds.  Used by the tests that pin "no memset node in
the step"  and by tools/graph_memset_probe.py.  torch's own ``CUDAGraph.debug_dump`` writes nothing
on this ROCm build, so the runtime is called directly on ``CUDAGraph(keep_graph=True).raw_cuda_graph()``."""
import ctypes as C

_KINDS = {0: "KERNEL", 1: "MEMCPY", 2: "MEMSET", 3: "HOST", 4: "GRAPH", 5: "EMPTY", 6: "WAIT_EVENT", 7: "EVENT_RECORD",
          8: "EXT_SEM_SIGNAL", 9: "EXT_SEM_WAIT", 10: "MEM_ALLOC", 11: "MEM_FREE", 12: "MEMCPY_FROM_SYMBOL", 13: "MEMCPY_TO_SYMBOL"}
_hip = None


def _rt():
    global _hip
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipGraphGetNodes.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _hip.hipGraphGetNodes.restype = C.c_int
        _hip.hipGraphNodeGetType.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
        _hip.hipGraphNodeGetType.restype = C.c_int
        _hip.hipGraphGetEdges.argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        _hip.hipGraphGetEdges.restype = C.c_int
        _hip.hipGraphDebugDotPrint.argtypes = [C.c_void_p, C.c_char_p, C.c_uint]
        _hip.hipGraphDebugDotPrint.restype = C.c_int
    return _hip


def node_kinds(raw_graph: int) -> dict:
    """-> {"KERNEL": n, "MEMSET": n, ..., "edges": n} of a hipGraph_t given as an integer handle."""
    hip = _rt()
    n = C.c_size_t(0)
    rc = hip.hipGraphGetNodes(C.c_void_p(raw_graph), None, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    nodes = (C.c_void_p * max(n.value, 1))()
    rc = hip.hipGraphGetNodes(C.c_void_p(raw_graph), nodes, C.byref(n))
    if rc != 0:
        raise RuntimeError(f"hipGraphGetNodes failed ({rc})")
    out = {}
    for i in range(n.value):
        t = C.c_int(-1)
        if hip.hipGraphNodeGetType(nodes[i], C.byref(t)) != 0:
            raise RuntimeError("hipGraphNodeGetType failed")
        k = _KINDS.get(t.value, f"TYPE_{t.value}")
        out[k] = out.get(k, 0) + 1
    e = C.c_size_t(0)
    if hip.hipGraphGetEdges(C.c_void_p(raw_graph), None, None, C.byref(e)) == 0:
        out["edges"] = int(e.value)
    out["nodes"] = int(n.value)
    return out


def dot_print(raw_graph: int, path: str, verbose: bool = False) -> bool:
    """hipGraphDebugDotPrint -> ``path``; True on success."""
    return _rt().hipGraphDebugDotPrint(C.c_void_p(raw_graph), path.encode(), 1 if verbose else 0) == 0
