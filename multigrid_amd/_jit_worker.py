"""hipRTC in a process of its own: `python _jit_worker.py <source.hip> <out.co> <option>...`.

Run as a script (never imported: importing the package would load PyTorch, whose wheel bundles its own -- possibly older --
libhiprtc / libamd_comgr, and a comgr that is already mapped wins over the ROCm installation's; seen: a bundled compiler without
__builtin_amdgcn_inverse_ballot_w64).  Exit code 0 = code object written; otherwise the compiler's log is on stderr."""
import ctypes as C
import os
import sys


def main() -> int:
    src_path, out_path, opts = sys.argv[1], sys.argv[2], sys.argv[3:]
    rtc = None
    for cand in (os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", "libhiprtc.so"), "libhiprtc.so"):
        try:
            rtc = C.CDLL(cand)
            break
        except OSError:
            continue
    if rtc is None:
        print("libhiprtc.so is not available", file=sys.stderr)
        return 3
    with open(src_path, "rb") as fh:
        src = fh.read()
    prog = C.c_void_p()
    rtc.hiprtcCreateProgram.argtypes = [C.POINTER(C.c_void_p), C.c_char_p, C.c_char_p, C.c_int, C.c_void_p, C.c_void_p]
    if rtc.hiprtcCreateProgram(C.byref(prog), src, b"mgx_jit.hip", 0, None, None) != 0:
        print("hiprtcCreateProgram failed", file=sys.stderr)
        return 4
    arr = (C.c_char_p * len(opts))(*[o.encode() for o in opts])
    rtc.hiprtcCompileProgram.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_char_p)]
    rc = rtc.hiprtcCompileProgram(prog, len(opts), arr)
    n = C.c_size_t()
    if rc != 0:
        rtc.hiprtcGetProgramLogSize(prog, C.byref(n))
        log = C.create_string_buffer(n.value + 1)
        rtc.hiprtcGetProgramLog(prog, log)
        sys.stderr.write(log.value.decode(errors="replace")[-6000:])
        return 5
    rtc.hiprtcGetCodeSize(prog, C.byref(n))
    code = C.create_string_buffer(n.value)
    rtc.hiprtcGetCode(prog, code)
    with open(out_path, "wb") as fh:
        fh.write(code.raw)
    return 0


if __name__ == "__main__":
    sys.exit(main())
