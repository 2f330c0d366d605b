import os, subprocess, sys
for sv in sys.argv[1:]:
    env = dict(os.environ, AFD_W33_S=sv)
    out = subprocess.run([sys.executable, "tools_conv_bench.py", "conv3p"], env=env, capture_output=True, text=True).stdout
    print("S", sv, out.strip()[-40:])
