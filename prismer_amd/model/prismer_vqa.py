"""VQA head (reference: model/prismer_vqa.py:15-122): question ‖ answer training loss with per-sample weights."""
import torch

from .prismer import Prismer
from .prismer_caption import tile, _first_token_probs


class PrismerVQA(Prismer):
    def _ids(self, text, device, **kw):
        if isinstance(text, dict):
            return text['input_ids'].to(device), text['attention_mask'].to(device)
        if isinstance(text, (tuple, list)) and len(text) == 2 and torch.is_tensor(text[0]):
            return text[0].to(device), text[1].to(device)
        t = self._tokenize(text, return_tensors='pt', **kw).to(device)
        return t.input_ids, t.attention_mask

    def forward(self, experts, question, answer=None, weights=None, train=True, inference='rank', k_test=128, return_scores=False):
        device = experts['rgb'].device
        pad = self.text_decoder.config.pad_token_id
        if isinstance(question, (list, tuple)) and question and isinstance(question[0], str):
            question = ['<s>' + q.capitalize() for q in question]
            q_ids, q_att = self._ids(question, device, padding='longest', truncation=True, max_length=35, add_special_tokens=False)
        else:
            q_ids, q_att = self._ids(question, device)
        if train:
            enc = self.expert_encoder(experts).permute(1, 0, 2)
            if isinstance(answer, (list, tuple)) and answer and isinstance(answer[0], str):
                answer = [' ' + a.capitalize() + '</s>' for a in answer]
                a_ids, a_att = self._ids(answer, device, padding='longest', add_special_tokens=False)
            else:
                a_ids, a_att = self._ids(answer, device)
            input_ids = torch.cat([q_ids, a_ids], dim=1).long()
            attention_mask = torch.cat([q_att, a_att], dim=1)
            targets = input_ids.masked_fill(input_ids == pad, -100)
            targets[:, :-a_ids.shape[1]] = -100                                           # prismer_vqa.py:32-33
            out = self.text_decoder(input_ids, attention_mask=attention_mask, encoder_hidden_states=enc, labels=targets, return_dict=True)
            return (weights.to(out.loss.dtype) * out.loss).mean()                           # prismer_vqa.py:40-41
        with torch.no_grad():
            enc = self.expert_encoder(experts).permute(1, 0, 2)
            if inference == 'generate':
                out = self.text_decoder.generate(input_ids=q_ids, encoder_hidden_states=enc, attention_mask=q_att,
                                                 max_length=q_ids.shape[1] + 10, min_length=q_ids.shape[1] + 2, num_beams=3,
                                                 length_penalty=-1)                     # prismer_vqa.py:52-58
                if self.tokenizer is None:
                    return out[:, q_ids.shape[1]:]
                return [self.tokenizer.decode(o[q_ids.shape[1]:], skip_special_tokens=True).lower().strip() for o in out]
            a_ids, a_att = self._ids(answer, device) if not (isinstance(answer, (list, tuple)) and isinstance(answer[0], str)) else \
                self._ids([' ' + a.capitalize() + '</s>' for a in answer], device, padding='longest', add_special_tokens=False)
            start = self.text_decoder(q_ids, attention_mask=q_att, encoder_hidden_states=enc, return_dict=True)
            prob_first = _first_token_probs(start.logits[:, -1, :], a_ids[:, 0])
            _, topk_ids = prob_first.topk(k_test, dim=1)
            ans_ids = torch.cat([a_ids.index_select(0, t) for t in topk_ids], dim=0)
            ans_att = torch.cat([a_att.index_select(0, t) for t in topk_ids], dim=0)
            input_ids = torch.cat([tile(q_ids, 0, k_test), ans_ids], dim=1).long()
            att = torch.cat([tile(q_att, 0, k_test), ans_att], dim=1)
            targets = input_ids.masked_fill(input_ids == pad, -100)
            targets[:, :-a_ids.shape[1]] = -100
            out = self.text_decoder(input_ids, attention_mask=att, encoder_hidden_states=tile(enc, 0, k_test), labels=targets, return_dict=True)
            lp = (-out.loss / torch.sum(targets != -100, dim=-1)).view(-1, k_test)
            best = lp.argmax(dim=1)
            if return_scores:                                   # (extension for the parity tests: the candidates and their scores)
                return topk_ids[best >= 0, best], topk_ids, lp
            return topk_ids[best >= 0, best]
