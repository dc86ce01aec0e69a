"""Stand-in for vamp_amd.do_vamp.fit_one used by the CPU test of the --parallel branch: records, from
inside the spawned worker, which file it was given, the GPU it was pinned to and whether the HIP
library had been loaded before the pin (it must not have been).  Test infrastructure."""
import json
import os
import sys


def record_fit(path, args, device=0):
    import vamp_amd
    rec = {"file": os.path.basename(path), "pid": os.getpid(), "hip_visible": os.environ.get("HIP_VISIBLE_DEVICES"),
           "device_arg": device, "lib_loaded_before_fit": vamp_amd._lib._lib is not None,
           "torch_imported": "torch" in sys.modules}
    with open(os.path.join(args.output_folder, os.path.basename(path) + ".json"), "w") as fh:
        json.dump(rec, fh)
