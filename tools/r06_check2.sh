# full GPU suite + smoke + bench lines of the headline, the video workloads and Swin-T (after the attention kernels' instruction diet)
cd ${GRAFT_REPO_ROOT:-.}
python3 -m pytest tests -q -m gpu 2>&1 | tail -3
python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
for wl in swin_b_w12_480_b2 video_swin_b_t8_384 video_swin_b_t8_384_sept swin_t_w7_480_b8; do
  python3 bench.py --no-cpu-baseline --no-profile --workload $wl 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$wl', d['ms_per_step'], d['value'])"
done
