cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { python3 bench.py --no-cpu-baseline --single-mode $2 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); k=d.get('kernels_ms') or {}; print('$1', d['ms_per_step'], {a: round(b, 4) for a, b in k.items() if isinstance(b, float)})"; }
run c2 "--steps 200 --warmup 20"
run p32 "--instances 32 --no-proxy-full --steps 300 --warmup 20"
run c4 "--arch PointNetPP --steps 40 --warmup 5 --presteps 20"
run c5 "--npoint 4096 --knn 32 --steps 30 --warmup 5 --presteps 40"
run c2cad "--data cad --steps 200 --warmup 20"
run c4cad "--data cad --arch PointNetPP --steps 40 --warmup 5 --presteps 20"
run c5cad "--data cad --npoint 4096 --knn 32 --steps 30 --warmup 5 --presteps 40"
