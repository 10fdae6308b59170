"""ctypes binding of libvirtex_amd.so (the C ABI declared in include/virtex_amd.h).

`import torch` happens first so that the HIP runtime already mapped by torch (its bundled
libamdhip64.so.7) is the one our library resolves against.

There is NO fallback: if the shared library is missing or a call fails, a RuntimeError is
raised.  The CPU fiber-emulator build of the same sources (tests/hipemu) can be selected
only explicitly, by tests, through `use_library()`.
"""
import ctypes
import os

import torch  # noqa: F401  (must precede loading the HIP library)

_HERE = os.path.dirname(os.path.abspath(__file__))
DEFAULT_LIB = os.path.join(_HERE, "lib", "libvirtex_amd.so")

_lib = None
_backend = None

c_void_p, c_int, c_float, c_u64, c_size_t = (ctypes.c_void_p, ctypes.c_int, ctypes.c_float,
                                             ctypes.c_uint64, ctypes.c_size_t)


class VtxError(RuntimeError):
    pass


def use_library(path: str):
    """Load a specific build (tests use this to select the hipemu build)."""
    global _lib, _backend
    if not os.path.exists(path):
        raise VtxError(
            f"virtex_amd native library not found at {path}; run `python -m virtex_amd.build` "
            "(the HIP extension is mandatory: there is no CPU/PyTorch fallback)")
    lib = ctypes.CDLL(path, mode=ctypes.RTLD_LOCAL)
    lib.vtx_last_error.restype = ctypes.c_char_p
    lib.vtx_backend.restype = ctypes.c_char_p
    _lib = lib
    _backend = lib.vtx_backend().decode()
    return lib


def lib():
    if _lib is None:
        use_library(os.environ.get("VIRTEX_AMD_LIB", DEFAULT_LIB))
    return _lib


def backend() -> str:
    lib()
    return _backend


def is_emulator() -> bool:
    return backend() == "hipemu"


_recorder = None     # virtex_amd.replay.Recorder while a step is being recorded (launch replay), else None


def set_recorder(rec):
    global _recorder
    _recorder = rec


def call(name: str, *args):
    """Invoke a C-ABI entry point; raise VtxError(vtx_last_error()) on non-zero status."""
    fn = getattr(lib(), name)
    rc = fn(*args)
    if rc != 0:
        raise VtxError(f"{name} failed ({rc}): {lib().vtx_last_error().decode()}")
    if _recorder is not None:            # the same call, re-issued by virtex_amd.replay (arguments are ctypes objects: kept as they are)
        def again(f=fn, a=args, n=name):
            if f(*a) != 0:               # a launch that fails during a replay raises exactly as the eager call does
                raise VtxError(f"{n} failed during launch replay: {lib().vtx_last_error().decode()}")
        _recorder.add("kernel", again, args, label=name)


# torch.cuda.current_stream() builds a Stream object through three Python layers (4.4 us per call, 2.5 ms of the 12 ms the
# host spends per step: tools/host_profile.py); the raw handle is one C call
_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream_handle(device) -> int:
    """hipStream_t (as an integer) of torch's current stream on `device`."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def stream_ptr(t: torch.Tensor):
    """hipStream_t the work for tensor `t` must be enqueued on."""
    if t.is_cuda:
        return c_void_p(current_stream_handle(t.device))
    if not is_emulator():
        raise VtxError("CPU tensor passed to the HIP build of virtex_amd (no CPU fallback exists)")
    return c_void_p(0)


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return c_void_p(0)
    if _recorder is not None:
        _recorder.keep.append(t)         # the recorded launch holds this address: the tensor must outlive the recording
    return c_void_p(t.data_ptr())


VTX_F32, VTX_BF16 = 0, 1


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return VTX_F32
    if dt == torch.bfloat16:
        return VTX_BF16
    raise VtxError(f"unsupported dtype {dt}")
