import torch
print("priority range (least, greatest):", torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream, 'priority_range') else None)
for p in (2, 1, 0, -1, -2):
    try:
        s = torch.cuda.Stream(priority=p); print("priority", p, "->", s.priority)
    except Exception as e:
        print("priority", p, "failed:", e)
