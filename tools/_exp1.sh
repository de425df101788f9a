cd $GRAFT_REPO_ROOT
python - <<'PY'
import subprocess, os, time, json, sys
sys.path.insert(0, '.')
from x265_amd.synth import make_clip
clip='/tmp/c.yuv'; make_clip(clip,1920,1080,120,seed=4321)
base=["--input",clip,"--input-res","1920x1080","--fps","30","--frames","120","--preset","medium","--me","hex","-o","/dev/null"]
def run(exe, extra, env=None):
    e=dict(os.environ, X265HIP_VERBOSE="1"); e.update(env or {})
    t=time.time(); p=subprocess.run([exe]+base+extra,capture_output=True,text=True,env=e); w=time.time()-t
    fps=[l for l in (p.stderr+p.stdout).splitlines() if l.startswith("encoded")]
    la=[l for l in p.stderr.splitlines() if "inside the seam" in l]
    return (fps[0].split("(")[1].split(")")[0] if fps else p.stderr[-200:], round(w,2), la[0][-40:] if la else "")
H="oracle/_ref/x265_hip_8bit"; R="oracle/_ref/x265_8bit"
out={}
for name,exe,extra,env in [("hip",H,[],None),("hip_pools64",H,["--pools","64"],None),("hip_pools32",H,["--pools","32"],None),("hip_pools16",H,["--pools","16"],None),
   ("hip_F8",H,["-F","8"],None),("hip_F3",H,["-F","3"],None),("hip_la10",H,["--rc-lookahead","10","--b-adapt","0"],None),("ref_la10",R,["--rc-lookahead","10","--b-adapt","0"],None),
   ("hip_pmode",H,["--pmode"],None),("ref_pmode",R,["--pmode"],None),("ref",R,[],None),("ref_pools32",R,["--pools","32"],None),("hip_bframes0",H,["--bframes","0"],None),("ref_bframes0",R,["--bframes","0"],None),
   ("hip_nowpp_F16",H,["--no-wpp","-F","16"],None)]:
    out[name]=run(exe,extra,env); print(name,out[name],flush=True)
json.dump(out,open("gpurun_out/exp1.json","w"),indent=1)
PY
