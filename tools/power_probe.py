"""Socket power and gfx clock of cuda:0 while a workload runs (verdict r4 item 4: is the SIREN step at the power cap?).

  python tools/power_probe.py OUT.txt TAG=cmd... [TAG=cmd ...]

Every `cmd` is run as a subprocess (shell) for as long as it takes; a sampler thread in THIS process reads, at >= 20 Hz,
  * amdsmi.amdsmi_get_gpu_metrics_info  (current_socket_power, average_socket_power, current_gfxclks per XCD,
    current_uclk, temperature_hotspot, throttle_status / indep_throttle_status, accumulated energy)
  * amdsmi.amdsmi_get_power_info / amdsmi_get_clock_info(GFX)  when the metrics table has no such field
  * sysfs hwmon power1_average / power1_input, freq1_input as a last resort
and the report lists, per workload, idle level, the samples inside the busy window (power above idle + 25 % of the swing),
their mean / p10 / p90, the power cap, and the violation (throttle) status the SMU reports.
A workload should keep the GPU busy for >= 2 s (tools/siren_loop.py, tools/probes/mfma_power SECONDS MODE ...).
"""
import glob
import os
import subprocess
import sys
import threading
import time


def _num(v):
    try:
        if v in ("N/A", None):
            return None
        return float(v)
    except (TypeError, ValueError):
        return None


class Sampler(threading.Thread):
    def __init__(self, hz=50):
        super().__init__(daemon=True)
        self.dt = 1.0 / hz
        self.samples = []          # (t, power_W, gfx_MHz_mean, gfx_MHz_max, uclk, hotspot_C, throttle, energy)
        self.stop_flag = False
        self.h = None
        self.smi = None
        self.info = {}
        try:
            import amdsmi
            amdsmi.amdsmi_init()
            hs = amdsmi.amdsmi_get_processor_handles()
            self.h, self.smi = hs[0], amdsmi
            try:
                self.info["power_cap"] = amdsmi.amdsmi_get_power_cap_info(self.h)
            except Exception as e:  # noqa: BLE001
                self.info["power_cap_err"] = repr(e)
            try:
                self.info["metrics_first"] = {k: v for k, v in amdsmi.amdsmi_get_gpu_metrics_info(self.h).items()
                                              if not isinstance(v, (list, dict)) or len(v) <= 16}
            except Exception as e:  # noqa: BLE001
                self.info["metrics_err"] = repr(e)
            try:
                self.info["power_info"] = amdsmi.amdsmi_get_power_info(self.h)
            except Exception as e:  # noqa: BLE001
                self.info["power_info_err"] = repr(e)
            try:
                self.info["clock_info"] = amdsmi.amdsmi_get_clock_info(self.h, amdsmi.AmdSmiClkType.GFX)
            except Exception as e:  # noqa: BLE001
                self.info["clock_info_err"] = repr(e)
        except Exception as e:  # noqa: BLE001
            self.info["amdsmi_err"] = repr(e)
        self.hw_power = (glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")
                         + glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
        self.hw_freq = glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input")
        self.info["sysfs"] = {"power": self.hw_power, "freq": self.hw_freq}

    def read(self):
        p = fm = fx = u = th = thr = en = None
        if self.h is not None:
            try:
                m = self.smi.amdsmi_get_gpu_metrics_info(self.h)
                p = _num(m.get("current_socket_power")) or _num(m.get("average_socket_power"))
                g = [x for x in (m.get("current_gfxclks") or []) if _num(x) not in (None, 0.0, 65535.0)]
                if g:
                    g = [float(x) for x in g]
                    fm, fx = sum(g) / len(g), max(g)
                elif _num(m.get("current_gfxclk")):
                    fm = fx = _num(m.get("current_gfxclk"))
                u = _num(m.get("current_uclk"))
                th = _num(m.get("temperature_hotspot"))
                thr = m.get("indep_throttle_status", m.get("throttle_status"))
                en = _num(m.get("energy_accumulator"))
            except Exception:  # noqa: BLE001
                pass
            if p is None:
                try:
                    pi = self.smi.amdsmi_get_power_info(self.h)
                    p = _num(pi.get("current_socket_power")) or _num(pi.get("socket_power")) or _num(pi.get("average_socket_power"))
                except Exception:  # noqa: BLE001
                    pass
            if fm is None:
                try:
                    ci = self.smi.amdsmi_get_clock_info(self.h, self.smi.AmdSmiClkType.GFX)
                    fm = fx = _num(ci.get("clk"))
                except Exception:  # noqa: BLE001
                    pass
        if p is None and self.hw_power:
            try:
                p = float(open(self.hw_power[0]).read()) / 1e6
            except Exception:  # noqa: BLE001
                pass
        if fm is None and self.hw_freq:
            try:
                fm = fx = float(open(self.hw_freq[0]).read()) / 1e6
            except Exception:  # noqa: BLE001
                pass
        return (time.time(), p, fm, fx, u, th, thr, en)

    def run(self):
        while not self.stop_flag:
            t0 = time.time()
            self.samples.append(self.read())
            d = self.dt - (time.time() - t0)
            if d > 0:
                time.sleep(d)


def stats(v):
    v = sorted(x for x in v if x is not None)
    if not v:
        return None
    n = len(v)
    return {"n": n, "mean": sum(v) / n, "p10": v[int(0.1 * (n - 1))], "p50": v[n // 2], "p90": v[int(0.9 * (n - 1))], "max": v[-1]}


def fmt(s, unit):
    if s is None:
        return "n/a"
    return "mean %.0f  p10 %.0f  p50 %.0f  p90 %.0f  max %.0f %s (n=%d)" % (s["mean"], s["p10"], s["p50"], s["p90"], s["max"], unit, s["n"])


def main():
    out = sys.argv[1]
    jobs = [a.split("=", 1) for a in sys.argv[2:]]
    s = Sampler()
    s.start()
    lines = ["# tools/power_probe.py: socket power / gfx clock of GPU 0 sampled at ~50 Hz through amdsmi while each workload runs",
             "# static info: %r" % (s.info,), ""]
    time.sleep(1.5)
    for tag, cmd in jobs:
        time.sleep(1.0)
        i0 = len(s.samples)
        t0 = time.time()
        r = subprocess.run(cmd, shell=True, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        t1 = time.time()
        time.sleep(0.5)
        win = s.samples[max(0, i0 - 25):]
        idle = [x[1] for x in s.samples[max(0, i0 - 25):i0] if x[1] is not None]
        idle_p = sum(idle) / len(idle) if idle else 0.0
        during = [x for x in win if t0 <= x[0] <= t1]
        pmax = max([x[1] for x in during if x[1] is not None] or [0.0])
        thr_p = idle_p + 0.25 * (pmax - idle_p)
        busy = [x for x in during if x[1] is not None and x[1] >= thr_p]
        lines.append("## %s   (%s)   wall %.1f s, rc %d" % (tag, cmd, t1 - t0, r.returncode))
        for l in (r.stdout or "").strip().splitlines()[-6:]:
            lines.append("   | " + l[:220])
        lines.append("   idle before: %.0f W;  busy window = samples with power >= %.0f W: %d of %d" % (idle_p, thr_p, len(busy), len(during)))
        lines.append("   socket power : " + fmt(stats([x[1] for x in busy]), "W"))
        lines.append("   gfxclk (mean over XCDs): " + fmt(stats([x[2] for x in busy]), "MHz"))
        lines.append("   gfxclk (max XCD)       : " + fmt(stats([x[3] for x in busy]), "MHz"))
        lines.append("   uclk: " + fmt(stats([x[4] for x in busy]), "MHz") + ";  hotspot: " + fmt(stats([x[5] for x in busy]), "C"))
        thr = sorted(set(repr(x[6]) for x in busy))
        lines.append("   throttle status values seen: %s" % (", ".join(thr[:6]),))
        en = [x for x in busy if x[7] is not None]
        if len(en) >= 2 and en[-1][0] > en[0][0]:
            lines.append("   energy accumulator: %.0f -> %.0f counts over %.2f s" % (en[0][7], en[-1][7], en[-1][0] - en[0][0]))
        if s.h is not None:
            try:
                lines.append("   violation status after: %r" % (s.smi.amdsmi_get_violation_status(s.h),))
            except Exception as e:  # noqa: BLE001
                lines.append("   violation status: %r" % (e,))
        # a decimated trace (every 5th sample of the run: ~10 Hz) so the time course is on record
        tr = ["%.2f:%sW/%sMHz" % (x[0] - t0, "%.0f" % x[1] if x[1] is not None else "?", "%.0f" % x[2] if x[2] is not None else "?")
              for x in during[::5]]
        for k in range(0, len(tr), 10):
            lines.append("   trace " + "  ".join(tr[k:k + 10]))
        lines.append("")
        print("\n".join(lines[-14 - (len(tr) + 9) // 10:]), flush=True)
    s.stop_flag = True
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "w").write("\n".join(lines) + "\n")


main()
