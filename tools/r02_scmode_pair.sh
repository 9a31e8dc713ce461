for L in tools/bin/libscpp_base.so scpp_amd/libscpp_hip.so tools/bin/libscpp_base.so scpp_amd/libscpp_hip.so; do
python bench.py --steps 1 --warmup 0 --no-cpu-baseline --library $PWD/$L 2>/dev/null | grep '^{' | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$L', 'headline', round(d['value'],1), 'sc_mode', round(d['config']['sc_mode']['terminated_trajectories_per_s'],1))"
done
timeout 300 python -m pytest tests -m gpu -x -q -k "twin or batch256 or converging or rocket2d" 2>&1 | tail -2
