"""Print what the SMI sources of bench.SmiSampler return on this box (raw amdsmi dictionaries, hwmon files) - run once on the GPU box."""
import glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
try:
    import amdsmi
    amdsmi.amdsmi_init()
    for h in amdsmi.amdsmi_get_processor_handles():
        print("bdf", amdsmi.amdsmi_get_gpu_device_bdf(h))
        print("power_info", amdsmi.amdsmi_get_power_info(h))
        print("power_cap_info", amdsmi.amdsmi_get_power_cap_info(h))
        print("clock_info GFX", amdsmi.amdsmi_get_clock_info(h, amdsmi.AmdSmiClkType.GFX))
except Exception as e:
    print("amdsmi:", type(e).__name__, e)
for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
    print(hw, os.path.realpath(os.path.join(hw, "..", "..")))
    for f in ("power1_average", "power1_input", "power1_cap", "freq1_input"):
        try:
            print("  ", f, open(os.path.join(hw, f)).read().strip())
        except Exception as e:
            print("  ", f, type(e).__name__)
import torch
import bench
s = bench.SmiSampler(0)
print("sampler source:", s.source, "cap", s.cap_w, "read:", s._read() if s._read else None)
