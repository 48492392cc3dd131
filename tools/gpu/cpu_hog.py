"""A CPU hog for the "busy box" runs of the GPU suite (tools/gpu/run.sh suite_hog): N processes spinning on integer work for a bounded time."""
import argparse
import multiprocessing as mp
import os
import signal
import time


def spin(seconds):
    t_end = time.time() + seconds
    x = 1
    while time.time() < t_end:
        for _ in range(200000):
            x = (x * 1103515245 + 12345) & 0x7FFFFFFF


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--fraction", type=float, default=0.5, help="of os.cpu_count() hardware threads")
    ap.add_argument("--seconds", type=float, default=600)
    a = ap.parse_args()
    n = max(1, int((os.cpu_count() or 1) * a.fraction))
    ps = [mp.Process(target=spin, args=(a.seconds,), daemon=True) for _ in range(n)]
    for p in ps:
        p.start()

    def stop(*_):  # `kill <this pid>` from run.sh: take the children along (daemon children only die with a clean exit)
        for p in ps:
            p.terminate()
        os._exit(0)

    signal.signal(signal.SIGTERM, stop)
    signal.signal(signal.SIGINT, stop)
    for p in ps:
        p.join()
