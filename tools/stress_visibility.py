"""Platform check: is a kernel's output always visible to the next kernel on the same stream while other streams are busy?"""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from foundationpose_cpp_amd import _lib
_lib.use_test_lib()
L = _lib.lib()
L.fpt_visibility_stress.restype = ctypes.c_longlong
for nth in (1, 2, 4):
    for mb in (10, 64):
        print(f"threads {nth} buffer {mb} MB: stale elements observed = {L.fpt_visibility_stress(nth, 3000, mb)}")
# the same check while the library's convolutions run on other streams
import threading
L.fpt_conv_stress.restype = ctypes.c_longlong
for shape in ((126, 40, 256, 256), (252, 20, 512, 512), (126, 40, 128, 128)):
    res = {}
    def load():
        res["conv"] = L.fpt_conv_stress(*shape, 1, 6000, 1, 0)
    t = threading.Thread(target=load); t.start()
    import time; time.sleep(5.0)
    stale = L.fpt_visibility_stress(2, 6000, 10)
    t.join()
    print(f"conv {shape} running alongside: stale elements = {stale}, conv mismatches = {res['conv']}")
