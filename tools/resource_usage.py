"""Resource remarks of the kernels of hlala_api.hip as one table (make -C hla-la_amd/csrc resource-usage 2>&1 | python tools/resource_usage.py):
registers, spills, scratch, occupancy, LDS per block."""
import re, sys, subprocess, shutil
rows = []; cur = {}
for l in sys.stdin:
    if 'remark:' not in l: continue
    t = l.split('remark:')[-1].strip()
    t = re.sub(r'\s*\[-Rpass-analysis=.*$', '', t)
    if 'Function Name' in t:
        if cur: rows.append(cur)
        cur = {'name': t.split('Function Name:')[1].strip()}
    elif ':' in t:
        k, v = t.split(':', 1); cur[k.strip()] = v.strip()
if cur: rows.append(cur)
filt = shutil.which('c++filt') or shutil.which('llvm-cxxfilt')
print("%-72s %5s %6s %6s %8s %4s %7s" % ("kernel", "VGPR", "vspill", "sspill", "scratch", "occ", "LDS"))
for r in rows:
    nm = r['name']
    if filt:
        nm = subprocess.run([filt, nm], capture_output=True, text=True).stdout.strip()
    nm = re.sub(r'^void ', '', nm); nm = re.sub(r'\((hlala::)?Dev.*$', '', nm); nm = nm.replace('hlala::', '')
    print("%-72s %5s %6s %6s %8s %4s %7s" % (nm[:72], r.get('VGPRs'), r.get('VGPRs Spill'), r.get('SGPRs Spill'), r.get('ScratchSize [bytes/lane]'), r.get('Occupancy [waves/SIMD]'), r.get('LDS Size [bytes/block]')))
