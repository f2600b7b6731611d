"""Host-side fixed costs on this box: clock reads, environment size, a ctypes call, torch stream / event operations."""
import ctypes, os, time, timeit
import torch
print("environment entries:", len(os.environ))
print("time.perf_counter(): %.3f us" % (timeit.timeit(time.perf_counter, number=200000) / 200000 * 1e6))
libc = ctypes.CDLL(None)
print("ctypes libc.getpid(): %.3f us" % (timeit.timeit(libc.getpid, number=100000) / 100000 * 1e6))
libc.getenv.restype = ctypes.c_char_p
print("ctypes getenv(absent): %.3f us" % (timeit.timeit(lambda: libc.getenv(b"ASVGP_NOT_SET"), number=100000) / 100000 * 1e6))
if torch.cuda.is_available():
    s = torch.cuda.Stream()
    e = torch.cuda.Event()
    e.record(s)
    torch.cuda.synchronize()
    print("event.query(): %.3f us" % (timeit.timeit(e.query, number=20000) / 20000 * 1e6))
    print("event.record(stream): %.3f us" % (timeit.timeit(lambda: e.record(s), number=5000) / 5000 * 1e6))
    torch.cuda.synchronize()
    print("stream.wait_event(): %.3f us" % (timeit.timeit(lambda: s.wait_event(e), number=5000) / 5000 * 1e6))
    torch.cuda.synchronize()
    def ctx():
        with torch.cuda.stream(s):
            pass
    print("with torch.cuda.stream(s): %.3f us" % (timeit.timeit(ctx, number=20000) / 20000 * 1e6))
    print("torch.cuda.current_stream().cuda_stream: %.3f us" % (timeit.timeit(lambda: torch.cuda.current_stream().cuda_stream, number=20000) / 20000 * 1e6))
    t = torch.zeros(8, dtype=torch.float64, device="cuda")
    print("tensor.data_ptr(): %.3f us" % (timeit.timeit(t.data_ptr, number=100000) / 100000 * 1e6))
    print("tensor[:4].tolist() (idle stream): %.3f us" % (timeit.timeit(lambda: t[:4].tolist(), number=2000) / 2000 * 1e6))
