"""Socket power and shader clock of THIS box's GPU while a command runs: python tools/power_sampler.py <label> -- <command ...>
Reads /sys/class/drm/card*/device/hwmon/*/power1_input (uW) and freq1_input (Hz) every 20 ms from a process that never touches HIP; the
card is the one whose PCI address `rocm-smi --showbus` reports for GPU[0] (the host's other cards belong to other boxes).  Prints one
JSON line: mean / median / p95 of power and clock over the samples in which the card was busy (power above idle + 150 W)."""
import glob, json, os, re, subprocess, sys, time


def my_card():
    try:
        out = subprocess.run(["rocm-smi", "--showbus"], capture_output=True, text=True, timeout=60).stdout
        m = re.search(r"GPU\[0\].*?([0-9a-fA-F]{4}:[0-9a-fA-F]{2}:[0-9a-fA-F]{2}\.[0-9a-fA-F])", out)
        bdf = m.group(1).lower() if m else None
    except Exception:
        bdf = None
    cards = []
    for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"):
        dev = os.path.realpath(os.path.join(h, "device"))
        cards.append((os.path.basename(dev).lower(), h))
    for b, h in cards:
        if bdf and b == bdf:
            return h, b
    return None, bdf


def read(path):
    with open(path) as f:
        return int(f.read().strip())


def main():
    label = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    h, bdf = my_card()
    all_h = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))
    watch = [h] if h else all_h
    idle = {w: read(os.path.join(w, "power1_input")) * 1e-6 for w in watch}
    child = subprocess.Popen(cmd)
    samples = {w: [] for w in watch}
    t0 = time.perf_counter()
    while child.poll() is None:
        for w in watch:
            try:
                samples[w].append((time.perf_counter() - t0, read(os.path.join(w, "power1_input")) * 1e-6, read(os.path.join(w, "freq1_input")) * 1e-6))
            except OSError:
                pass
        time.sleep(0.02)
    if not h:        # no bus id: the card whose power moved most is ours
        h = max(watch, key=lambda w: max((s[1] for s in samples[w]), default=0) - idle[w])
    s = samples[h]
    busy = [x for x in s if x[1] > idle[h] + 150.0]

    def stats(v):
        v = sorted(v)
        if not v:
            return None
        return dict(mean=round(sum(v) / len(v), 1), median=round(v[len(v) // 2], 1), p95=round(v[int(len(v) * 0.95)], 1), max=round(v[-1], 1))
    print(json.dumps(dict(label=label, card=h, bus=bdf, idle_w=round(idle[h], 1), cap_w=read(os.path.join(h, "power1_cap")) * 1e-6, samples=len(s),
                          busy_samples=len(busy), power_w=stats([x[1] for x in busy]), sclk_mhz=stats([x[2] for x in busy]), exit=child.returncode)), flush=True)
    if os.environ.get("POWER_SERIES"):
        with open(os.environ["POWER_SERIES"], "w") as f:
            for x in s:
                f.write(f"{x[0]:.3f} {x[1]:.0f} {x[2]:.0f}\n")
    sys.exit(child.returncode)


if __name__ == "__main__":
    main()
