GATHER_MIX_ZIPF=0.6,0.8,0.9,1.0,1.1,1.2 timeout -k 10 200 bin/gather_mix 512 0 2097152 0 0
