# round 4: board power and shader clock while the DFPN / CHN steps replay (is the step at the power cap?).
# The parent never touches the GPU: it starts bench.py as a child and polls sysfs / rocm-smi.
import glob, json, os, subprocess, sys, time
def read(p):
    try: return open(p).read().strip()
    except Exception: return None
hw = sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*'))
print('hwmon dirs:', hw, flush=True)
for h in hw:
    for f in sorted(os.listdir(h)):
        if f.startswith(('power', 'freq', 'temp1_input', 'name')):
            print(' ', h, f, read(os.path.join(h, f)), flush=True)
def smi():
    try:
        out = subprocess.run(['rocm-smi', '--showpower', '--showclocks', '--showmaxpower', '--json'], capture_output=True, text=True, timeout=20).stdout
        return json.loads(out)
    except Exception as e:
        return {'error': repr(e)}
print('idle:', json.dumps(smi())[:1500], flush=True)
for only, tag in (('--no-chn', 'dfpn'), ('--no-dfpn', 'chn')):
    child = subprocess.Popen([sys.executable, 'bench.py', '--steps', '150', '--warmup', '5', '--no-cpu-baseline', '--no-check', only],
                             stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    samples = []
    t0 = time.time()
    while child.poll() is None and time.time() - t0 < 240:
        s = {'t': round(time.time() - t0, 1)}
        for h in hw:
            for f in ('power1_average', 'power1_input', 'freq1_input'):
                v = read(os.path.join(h, f))
                if v is not None: s[f] = int(v)
        if not any(k.startswith('power') for k in s):
            j = smi()
            c = j.get('card0', {}) if isinstance(j, dict) else {}
            for k, v in c.items():
                if 'Power' in k or 'sclk' in k: s[k] = v
        samples.append(s)
        time.sleep(0.25)
    line = child.stdout.read().strip().splitlines()
    d = json.loads(line[-1]) if line else {}
    ms = d.get('ms_per_step') if tag == 'dfpn' else (d.get('chn_train_step') or {}).get('ms_per_step')
    print(tag, 'ms_per_step', ms, 'samples', len(samples), flush=True)
    for s in samples: print(' ', tag, json.dumps(s), flush=True)
print('after:', json.dumps(smi())[:1500], flush=True)
