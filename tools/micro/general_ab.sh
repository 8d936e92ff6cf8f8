# the general kernel on its workloads: config E (stereo ASRC, 65,536-frame calls), config P pinned, the headline shape pinned, a stereo
# interpolating ASRC at 16,384 frames, 8-byte samples — end-to-end rate per case (tools/profile_case.py prints it)
for c in general_E general_P general_A strict; do python tools/profile_case.py $c 60 2>&1 | grep -v amdgpu.ids | python -c "
import sys,json
for l in sys.stdin:
    try: d=json.loads(l); print(d.get('case'), {k:d[k] for k in d if k in ('Msamples_per_s','ms_per_step','kernel')})
    except Exception: print(l.rstrip())
"; done
for sh in "2 380 380 44100 48001 0 1 16384 1" "2 380 380 44100 48001 0 0 65536 1" "8 988 988 44100 48001 0 1 65536 1" "1 48 48 44100 48001 0 1 65536 1" "2 156 156 44100 48001 0 1 4096 1"; do python tools/bench_shapes.py $sh 2>&1 | grep -v amdgpu.ids; done
