"""Start the HIP runtime on a background thread before anything heavy is imported (command-line launchers only).

Creating the GPU context costs a few hundred milliseconds, about as long as `import numpy`; the two run side by side
when the launcher calls start() first.  Nothing here is required for correctness: the library initialises itself on
first use whether or not this ran."""
import ctypes
import os
import threading


def start():
    def run():
        try:
            hip = ctypes.CDLL("libamdhip64.so")
            hip.hipInit(0)
            hip.hipSetDevice(int(os.environ.get("SK_DEVICE", os.environ.get("LOCAL_RANK", "0")) or 0))
            hip.hipFree(None)                                    # forces the primary context into existence
        except Exception:                                        # noqa: BLE001 -- no runtime here: first use reports it
            pass
    t = threading.Thread(target=run, name="sk-hip-warm", daemon=True)
    t.start()
    return t
