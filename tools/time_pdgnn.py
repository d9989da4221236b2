"""PDGNN forward vs exact PD on HIV-shaped molecules (bench.py's auxiliary block on its own) -- development aid."""
import sys
import torch
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import bench
print(bench.pdgnn_aux(torch, torch.device("cuda:0")))
