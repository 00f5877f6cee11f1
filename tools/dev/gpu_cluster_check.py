import sys
sys.path.insert(0, ".")
from tests.gpu_check import check_run
check_run("rastrigin", 2, 0, 300, 6, 1, -5.12, 5.12, seed=2, clustering=1)
check_run("rastrigin", 2, 0, 300, 6, 40, -5.12, 5.12, seed=2, clustering=1)
check_run("twin_gaussian", 6, 1, 150, 12, 30, -1.0, 1.0, seed=4, clustering=1)
check_run("rastrigin", 4, 0, 200, 12, 50, -5.12, 5.12, seed=5, clustering=1)
