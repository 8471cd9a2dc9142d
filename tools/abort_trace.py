"""Diagnostics: run pytest in-process with a SIGABRT handler that prints the Python stacks and the native backtrace (glibc backtrace_symbols_fd)."""
import ctypes, faulthandler, signal, sys
libc = ctypes.CDLL("libc.so.6")
HANDLER = ctypes.CFUNCTYPE(None, ctypes.c_int)
def _on_abort(sig):          # called from the C signal handler itself (not deferred to the interpreter loop, which is gone at exit)
    buf = (ctypes.c_void_p * 64)()
    n = libc.backtrace(buf, 64)
    libc.backtrace_symbols_fd(buf, n, 2)
_cb = HANDLER(_on_abort)
libc.signal(signal.SIGABRT, _cb)
import pytest
rc = pytest.main(sys.argv[1:])
print("pytest rc", rc, flush=True)
