mkdir -p /tmp/w
loop() { tag=$1; shift; for i in $(seq 1 30); do env "$@" X265HIP=require X265HIP_VERBOSE=1 TWO_ENCODERS_WATCHDOG=20 timeout 60 oracle/_ref/two_encoders_hip8 /tmp/w/w par 2> gpurun_out/hang_$tag.log; rc=$?; if [ $rc -ne 0 ]; then echo "$tag: iteration $i rc=$rc"; return; fi; done; echo "$tag: 30 clean"; }
loop all A=1
loop nosrc X265HIP_SRCPLANES=0
loop noref X265HIP_REFPLANES=0
loop nola X265HIP_LOOKAHEAD=0
