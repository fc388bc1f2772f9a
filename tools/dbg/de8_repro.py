import os, sys, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, REPO)
from test_narrow_gpu import check_de8_stack
gpu = torch.device("cuda", 0)
variant, N, B, bf16, train = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4] == "1", sys.argv[5] == "1"
for Ly in (1, 2, 3):
    try:
        check_de8_stack(variant, N, 64, bf16, train, gpu, B=B, Ly=Ly); print("Ly", Ly, "ok")
    except AssertionError as e:
        print("Ly", Ly, "FAIL", str(e)[:250])
