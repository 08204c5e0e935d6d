"""CPU restatement of the three train_epoch / validate pairs (the measured unit)."""
import torch

from .losses import concordance_index_np, cox_loss, gate_entropy_loss, neg_partial_log_likelihood


def train_epoch_final(model, loader, optimizer, device, on_batch=None):
    """final_multimodal.py:238-265: Cox on the whole batch, no label mask.  (on_batch: test hook, called with each batch's loss.)"""
    model.train()
    total, nb = 0.0, 0
    for batch in loader:
        ct, rna, clin = batch['image'].to(device), batch['rnaseq'].to(device), batch['clinical'].to(device)
        label = batch['label'].to(device)
        time, event = label[:, 0], label[:, 1]
        hazard = model(ct, rna, clin)
        loss = cox_loss(hazard, event, time)
        optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        total += loss.item()
        nb += 1
        if on_batch is not None:
            on_batch(loss.item())
    return total / nb if nb > 0 else 0


def validate_final(model, loader, device, tie_credit=0.5):
    """final_multimodal.py:268-305 (C-index: torchsurv / lifelines rule = 0.5 credit for tied scores)."""
    model.eval()
    total, nb = 0.0, 0
    hs, ts, es = [], [], []
    with torch.no_grad():
        for batch in loader:
            ct, rna, clin = batch['image'].to(device), batch['rnaseq'].to(device), batch['clinical'].to(device)
            label = batch['label'].to(device)
            time, event = label[:, 0], label[:, 1]
            hazard = model(ct, rna, clin)
            total += cox_loss(hazard, event, time).item()
            nb += 1
            hs.extend(hazard.cpu().numpy()); ts.extend(time.cpu().numpy()); es.extend(event.cpu().numpy())
    return (total / nb if nb > 0 else 0), concordance_index_np(hs, es, ts, tie_credit=tie_credit)


def train_epoch_partial(model, loader, optimizer, device, gate_entropy_weight=0.01, on_batch=None):
    """partial_modality_training.py:382-435.  (on_batch: test hook, called with each batch's (Cox, entropy) losses.)"""
    model.train()
    tot_cox, tot_ent, n_surv, nb = 0.0, 0.0, 0, 0
    for batch in loader:
        ct, rna, clin = batch['image'].to(device), batch['rnaseq'].to(device), batch['clinical'].to(device)
        label, mask = batch['label'].to(device), batch['mask'].to(device)
        hazard, gate = model(ct, rna, clin, mask)
        smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
        c_loss = torch.tensor(0.0, device=device)
        if smask.sum() > 0:
            h, t, e = hazard[smask], label[smask, 0], label[smask, 1]
            if h.shape[0] >= 2 and e.sum() > 0:
                c_loss = cox_loss(h, e, t)
                tot_cox += c_loss.item()
                n_surv += 1
        e_loss = gate_entropy_loss(gate)
        tot_ent += e_loss.item()
        loss = c_loss + gate_entropy_weight * e_loss
        optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        nb += 1
        if on_batch is not None:
            on_batch(float(c_loss.item()), e_loss.item())
    return (tot_cox / n_surv if n_surv > 0 else 0), (tot_ent / nb if nb > 0 else 0)


def train_epoch_simple(model, loader, optimizer, device, on_batch=None):
    """simple_fusion.py:242-279 (both `continue`s kept, the second one after the forward).  (on_batch: test hook, each stepped batch's loss.)"""
    model.train()
    total, nb = 0.0, 0
    for batch in loader:
        image, rnaseq = batch['image'].to(device), batch['rnaseq'].to(device)
        time, event = batch['time'].squeeze().to(device), batch['event'].squeeze().to(device)
        smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
        if smask.sum() < 2:
            continue
        optimizer.zero_grad()
        lh = model(image, rnaseq)
        lh_s, t_s, e_s = lh[smask], time[smask], event[smask].bool()
        if e_s.sum() == 0:
            continue
        loss = neg_partial_log_likelihood(lh_s, e_s, t_s)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 1.0)
        optimizer.step()
        total += loss.item()
        nb += 1
        if on_batch is not None:
            on_batch(loss.item())
    return total / nb if nb > 0 else 0.0


def train_epoch_rnaseq(model, loader, optimizer, device="cpu"):
    """train_rnaseq_only.py:157-176: NPLL on the whole batch, NO gradient clipping, mean over len(loader)."""
    model.train()
    total = 0.0
    for batch in loader:
        rnaseq = batch['rnaseq'].to(device)
        time, event = batch['time'].squeeze().to(device), batch['event'].squeeze().to(device)
        optimizer.zero_grad()
        loss = neg_partial_log_likelihood(model(rnaseq).squeeze(), event, time)
        loss.backward()
        optimizer.step()
        total += loss.item()
    return total / len(loader)


def validate_rnaseq(model, loader, device="cpu"):
    """train_rnaseq_only.py:178-209."""
    model.eval()
    total, hs, ts, es = 0.0, [], [], []
    with torch.no_grad():
        for batch in loader:
            rnaseq = batch['rnaseq'].to(device)
            time, event = batch['time'].squeeze().to(device), batch['event'].squeeze().to(device)
            hz = model(rnaseq).squeeze()
            total += neg_partial_log_likelihood(hz, event, time).item()
            hs.extend(hz.cpu().numpy()); ts.extend(time.cpu().numpy()); es.extend(event.cpu().numpy())
    return total / len(loader), concordance_index_np(hs, es, ts)


def validate_partial(model, loader, device, tie_credit=0.5):
    """partial_modality_training.py:438-485: eval forward of every batch; only batches with >= 2 labelled patients and >= 1
    event contribute a loss term AND their labelled hazards to the C-index.  C-index rule: calculate_cindex = torchsurv /
    lifelines (0.5 credit for tied scores; tie_credit=0 = the simple_fusion fallback counting)."""
    model.eval()
    total, nb = 0.0, 0
    hs, ts, es = [], [], []
    with torch.no_grad():
        for batch in loader:
            ct, rna, clin = batch['image'].to(device), batch['rnaseq'].to(device), batch['clinical'].to(device)
            label, mask = batch['label'].to(device), batch['mask'].to(device)
            hazard, _ = model(ct, rna, clin, mask)
            smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
            if smask.sum() > 0:
                h, t, e = hazard[smask], label[smask, 0], label[smask, 1]
                if h.shape[0] >= 2 and e.sum() > 0:
                    total += cox_loss(h, e, t).item()
                    nb += 1
                    hs.extend(h.cpu().numpy()); ts.extend(t.cpu().numpy()); es.extend(e.cpu().numpy())
    c = concordance_index_np(hs, es, ts, tie_credit=tie_credit) if hs else 0.5
    return (total / nb if nb > 0 else 0), c


def validate_simple(model, loader, device, tie_credit=0.0):
    """simple_fusion.py:281-333: batches with < 2 labelled patients are skipped BEFORE the forward, batches without events after it."""
    model.eval()
    total, nb = 0.0, 0
    hs, ts, es = [], [], []
    with torch.no_grad():
        for batch in loader:
            image, rnaseq = batch['image'].to(device), batch['rnaseq'].to(device)
            time, event = batch['time'].squeeze().to(device), batch['event'].squeeze().to(device)
            smask = torch.as_tensor(batch['has_survival'], dtype=torch.bool, device=device)
            if smask.sum() < 2:
                continue
            lh = model(image, rnaseq)
            lh_s, t_s, e_s = lh[smask], time[smask], event[smask].bool()
            if e_s.sum() == 0:
                continue
            total += neg_partial_log_likelihood(lh_s, e_s, t_s).item()
            nb += 1
            hs.append(lh_s.cpu()); ts.append(t_s.cpu()); es.append(e_s.cpu())
    if not hs:
        return 0.0, 0.5
    c = concordance_index_np(torch.cat(hs).numpy(), torch.cat(es).numpy(), torch.cat(ts).numpy(), tie_credit=tie_credit)
    return (total / nb if nb > 0 else 0.0), c
