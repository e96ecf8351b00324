# calibration of rocprofv3 FETCH_SIZE on gfx950 for the two access widths the hot kernels use: a known-size streaming read
# (4 GiB, beyond the 256 MiB Infinity Cache) with 8 and 16 bytes per lane; prints counted / actual bytes.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/calib.py <<'PY'
import sys, torch
sys.path.insert(0, '/root/repo')
from infgen_amd import _lib
lib = _lib.load()
n = 4 << 30
buf = torch.empty(n // 4, device='cuda', dtype=torch.float32).normal_()
out = torch.zeros(2048, device='cuda')
st = torch.cuda.current_stream().cuda_stream
for w in (8, 16, 8, 16):
    _lib.check(lib.infgen_debug_stream_read(buf.data_ptr(), n, w, out.data_ptr(), st))
torch.cuda.synchronize()
PY
rm -rf /tmp/calib
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/calib -- python /tmp/calib.py > /tmp/calib.log 2>&1
f=$(find /tmp/calib -name "*counter_collection.csv" | head -1)
python - "$f" <<'PY'
import csv, sys
n = 4 << 30
for r in csv.DictReader(open(sys.argv[1])):
    if 'k_stream_read' in r['Kernel_Name'] and r['Counter_Name'] == 'FETCH_SIZE':
        kb = float(r['Counter_Value'])
        print(f"{r['Kernel_Name'][:40]:40s} FETCH_SIZE {kb:.0f} KB  counted/actual = {kb * 1024 / n:.4f}")
PY
