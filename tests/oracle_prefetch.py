"""Background evaluation of the full-geometry oracle cases (TEST INFRASTRUCTURE).

tests/test_full_geometry_gpu.py compares the engine with the CPU oracle at Flux-dev's real widths.  Per case the host side --
synthesising the checkpoint (the 19 + 38-block one is 11.9 B uniform draws from ONE sequential CPU generator: ~3 minutes on a single
core, and its bytes are pinned by the fixtures, so it cannot be generated any other way) and the oracle's calibrating + frozen calls --
took 60-80 % of the GPU suite's 12.7 minutes while the GPU idled (profiles/r05_pytest_gpu_durations.txt).  None of it depends on the
GPU, so worker threads start on it when the session starts (tests/conftest.py schedules the cases the collected tests will ask for, in
test order; the full-depth checkpoint gets a thread of its own) and `take(name)` hands a finished case to the test, or waits for it.
torch's CPU kernels release the GIL, so the workers overlap with the GPU tests of the main thread.  A case that was not scheduled (a
single test selected by hand) is computed by the caller as before: same code, same results -- `compute` is the one implementation.
"""
import threading
import time
import traceback

_jobs = {}
_lock = threading.Lock()
_threads = []
_cancel = threading.Event()


class Cancelled(BaseException):
    """raised inside a worker (from its log callback, the only cancellation points the oracle code offers) once the session is over"""


class _Job:
    def __init__(self):
        self.done = threading.Event()
        self.result = None
        self.error = None
        self.log = []


def compute(name, log):
    """-> (case, p, sd, inp, orc, pred_calib, pred_frozen, trace): what prepare_case needs from the host side"""
    import full_geometry as fg
    from fluxmi import synth

    t0 = time.time()
    case, p, sd, inp = fg.make_case(name, synth)
    log(f"[{name}] synthetic checkpoint {sum(v.numel() for v in sd.values()) / 1e9:.2f} B parameters in {time.time() - t0:.0f} s")
    orc, o0, o1, tr = fg.run_oracle(name, p, sd, inp, log=log)
    return case, p, sd, inp, orc, o0, o1, tr


def _worker(work):
    import torch

    torch.set_num_threads(torch.get_num_threads())  # a new thread starts from the OpenMP default: give it the session's team size
    def log(line, _append=None):
        if _cancel.is_set():
            raise Cancelled()
        _append(line)

    for name, job in work:  # the job objects themselves: take() may already have removed a name from the table
        if _cancel.is_set():
            job.error = "cancelled: the session ended before this case was computed"
            job.done.set()
            continue
        try:
            job.result = compute(name, lambda line, a=job.log.append: log(line, a))
        except BaseException:  # noqa: the test that takes the case re-raises
            job.error = traceback.format_exc()
        job.done.set()


def schedule(groups):
    """groups: lists of case names, one worker thread per list (cases of a list run in order)"""
    with _lock:
        for names in groups:
            names = [n for n in names if n not in _jobs]
            if not names:
                continue
            for n in names:
                _jobs[n] = _Job()
            t = threading.Thread(target=_worker, args=([(n, _jobs[n]) for n in names],), daemon=True, name="oracle-prefetch:" + names[0])
            _threads.append(t)
            t.start()


def cancel_and_join(timeout: float = 20.0) -> bool:
    """End of session: ask the workers to stop at their next log line and wait for them.  -> True when no worker is left running.
    A daemon thread still inside torch's C++ when the interpreter finalises is killed by pthread_exit while unwinding through noexcept
    frames -> std::terminate -> the whole pytest process aborts with rc 134 whatever the tests did (GPUTEST_r05); the caller (conftest)
    leaves through os._exit when this returns False."""
    _cancel.set()
    t_end = time.time() + timeout
    for t in _threads:
        t.join(max(0.0, t_end - time.time()))
    return not any(t.is_alive() for t in _threads)


def take(name):
    """the finished case (waits for it), or None when nobody scheduled it.  A case is handed out once (its tensors are GBs)."""
    with _lock:
        job = _jobs.pop(name, None)
    if job is None:
        return None
    t0 = time.time()
    job.done.wait()
    if job.error:
        raise RuntimeError(f"background oracle evaluation of {name} failed:\n{job.error}")
    for line in job.log:
        print(line, flush=True)
    print(f"[{name}] host side came from the background worker (waited {time.time() - t0:.0f} s for it)", flush=True)
    return job.result
