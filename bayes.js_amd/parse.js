'use strict';
/*
 * parse.js -- the front half of the log_post translator (translate.js): tokenizer, recursive-descent parser for the numeric
 * JavaScript subset, the rewriting of post-ES5 spellings (destructuring, for-of, forEach / reduce / map, Array(n).fill(v), switch,
 * do-while, helpers that are handed objects) into the core subset, and the AST walks the translator's analyses use.
 * AST nodes are plain objects {k: kind, ...}:
 *   statements   Block{body} VarDecl{kind, decls:[{name, init}]} ExprStmt{expr} If{test, cons, alt} For{init, test, update, body}
 *                (while and do-while are For nodes) Return{arg} Break Continue Empty
 *   expressions  Num{v} Str{v} Bool{v} Id{name} Member{obj, prop} Index{obj, idx} Call{callee, args} Unary{op, arg}
 *                Binary{op, l, r} Logical{op, l, r} Cond{test, a, b} Assign{op, target, value} Update{op, prefix, target}
 *                Seq{l, r} ArrayLit{elems} Func{params, body} NewArray{len, fill}
 * Order of the passes in parseFunctionSource: parse (destructuring, for-of, switch, do-while are rewritten on the fly) ->
 * desugarBlock (forEach / reduce / map / Array(n) hoisting) -> rewritePushLoops -> uniquifyBlockScoped.  The translator runs
 * desugarBlock once more with P.env set, to inline helpers that are handed the state or the data.
 */
// ------------------------------------------------------------------------------------------
// tokenizer
const PUNCT = ['===', '!==', '>>>', '**', '==', '!=', '<=', '>=', '&&', '||', '++', '--', '+=', '-=', '*=', '/=', '%=', '=>', '<<', '>>',
  '{', '}', '(', ')', '[', ']', ';', ',', '.', '?', ':', '<', '>', '+', '-', '*', '/', '%', '!', '=', '|', '&', '^', '~'];

function tokenize(src) {
  const out = [];
  let i = 0, nl = false;
  const n = src.length;
  while (i < n) {
    const c = src[i];
    if (c === '\n') { nl = true; i++; continue; }
    if (c === ' ' || c === '\t' || c === '\r') { i++; continue; }
    if (c === '/' && src[i + 1] === '/') { while (i < n && src[i] !== '\n') i++; continue; }
    if (c === '/' && src[i + 1] === '*') {
      const e = src.indexOf('*/', i + 2);
      if (e < 0) throw 'unterminated comment';
      if (src.slice(i, e).indexOf('\n') >= 0) nl = true;
      i = e + 2; continue;
    }
    if (/\d/.test(c) || (c === '.' && /\d/.test(src[i + 1] || ''))) {
      const m = /^(?:0[xX][0-9a-fA-F]+|(?:\d+\.?\d*|\.\d+)(?:[eE][+-]?\d+)?)/.exec(src.slice(i, i + 64));
      out.push({ t: 'num', v: Number(m[0]), nl }); nl = false; i += m[0].length; continue;
    }
    if (/[A-Za-z_$]/.test(c)) {
      let j = i + 1;
      while (j < n && /[\w$]/.test(src[j])) j++;
      out.push({ t: 'id', v: src.slice(i, j), nl }); nl = false; i = j; continue;
    }
    if (c === '"' || c === "'") {
      let j = i + 1, s = '';
      while (j < n && src[j] !== c) { if (src[j] === '\\') { j++; } s += src[j]; j++; }
      if (j >= n) throw 'unterminated string literal';
      out.push({ t: 'str', v: s, nl }); nl = false; i = j + 1; continue;
    }
    let hit = null;
    for (const p of PUNCT) if (src.startsWith(p, i)) { hit = p; break; }
    if (!hit) throw 'log_post uses a character this translator does not understand: ' + JSON.stringify(c);
    out.push({ t: 'p', v: hit, nl }); nl = false; i += hit.length;
  }
  out.push({ t: 'eof', v: '<end>', nl: true });
  return out;
}

// ------------------------------------------------------------------------------------------
// parser (recursive descent, JavaScript precedence)
const KEYWORDS = new Set(['var', 'let', 'const', 'for', 'while', 'if', 'else', 'return', 'function', 'true', 'false',
  'break', 'continue', 'do', 'switch', 'case', 'default', 'new', 'typeof', 'in', 'of', 'this', 'null', 'undefined', 'throw', 'try']);

function Parser(tokens) { this.tk = tokens; this.i = 0; }
Parser.prototype = {
  peek(v) { const t = this.tk[this.i]; return t.t !== 'num' && t.t !== 'str' && t.v === v; },
  peekAt(k, v) { const t = this.tk[this.i + k]; return t && t.t !== 'num' && t.t !== 'str' && t.v === v; },
  eat(v) { if (this.peek(v)) { this.i++; return true; } return false; },
  expect(v) { if (!this.eat(v)) throw "expected '" + v + "' but found '" + this.tk[this.i].v + "' in log_post"; },
  ident() { const t = this.tk[this.i]; if (t.t !== 'id' || KEYWORDS.has(t.v)) throw "expected a name but found '" + t.v + "'"; this.i++; return t.v; },
  endStmt() {   // ';' or automatic semicolon insertion (before '}', at a line break, at the end)
    if (this.eat(';')) return;
    const t = this.tk[this.i];
    if (t.t === 'eof' || this.peek('}') || t.nl) return;
    throw "expected ';' but found '" + t.v + "' in log_post";
  },
  fresh(stem) { this.uniq = (this.uniq || 0) + 1; return '__' + stem + this.uniq; },
  // a binding pattern: a name, {a, b: c, d: {e}} or [a, , b] (no defaults, no rest) -> {name} | {props: [[key, pattern]]} | {elems: [pattern|null]}
  pattern() {
    if (this.eat('{')) {
      const props = [];
      if (!this.peek('}')) do {
        if (this.peek('}')) break;
        const t = this.tk[this.i];
        if (t.t !== 'id' && t.t !== 'str') throw "expected a property name in a destructuring pattern but found '" + t.v + "'";
        this.i++;
        if (this.eat(':')) props.push([t.v, this.pattern()]);
        else { if (KEYWORDS.has(t.v)) throw "expected a name but found '" + t.v + "'"; props.push([t.v, { name: t.v }]); }
        if (this.peek('=')) throw 'default values in destructuring patterns are not supported';
      } while (this.eat(','));
      this.expect('}');
      return { props };
    }
    if (this.eat('[')) {
      const elems = [];
      while (!this.peek(']')) {
        if (this.peek(',')) { this.i++; elems.push(null); continue; }
        if (this.peek('.')) throw 'rest elements in destructuring patterns are not supported';
        elems.push(this.pattern());
        if (this.peek('=')) throw 'default values in destructuring patterns are not supported';
        if (!this.peek(']')) this.expect(',');
      }
      this.expect(']');
      return { elems };
    }
    return { name: this.ident() };
  },
  // declarations binding `pat` to the (side-effect free) expression `src`
  bind(pat, src, decls) {
    if (pat.name) { decls.push({ name: pat.name, init: src }); return; }
    if (pat.props) for (const [key, sub] of pat.props) this.bind(sub, { k: 'Member', obj: src, prop: key }, decls);
    else pat.elems.forEach((sub, i) => { if (sub) this.bind(sub, { k: 'Index', obj: src, idx: { k: 'Num', v: i } }, decls); });
  },
  // parameter list -> names; destructured parameters get a synthetic name and declarations that go in front of the body
  paramList(prologue) {
    const params = [];
    if (!this.peek(')')) do {
      const pat = this.pattern();
      if (pat.name) params.push(pat.name);
      else { const nm = this.fresh('arg'); params.push(nm); const decls = []; this.bind(pat, { k: 'Id', name: nm }, decls); prologue.push({ k: 'VarDecl', kind: 'var', decls }); }
      if (this.peek('=')) throw 'default parameter values are not supported';
    } while (this.eat(','));
    return params;
  },
  functionBody(arrow, prologue) {
    let body;
    if (this.peek('{')) body = this.block();
    else if (arrow) body = { k: 'Block', body: [{ k: 'Return', arg: this.assignment() }] };
    else throw 'log_post must be a function expression or an arrow function';
    if (prologue.length) body = { k: 'Block', body: prologue.concat(body.body) };
    return body;
  },
  parseFunction() {
    let params = [];
    const prologue = [];
    if (this.eat('function')) { if (!this.peek('(')) this.ident(); }
    if (this.eat('(')) {
      params = this.paramList(prologue);
      this.expect(')');
    } else {
      params.push(this.ident());    // x => ...
    }
    const arrow = this.eat('=>');
    const body = this.functionBody(arrow, prologue);
    if (this.tk[this.i].t !== 'eof') throw "unexpected '" + this.tk[this.i].v + "' after the end of the function";
    return { params, body: desugarBlock(body, this) };
  },
  block() { this.expect('{'); const body = []; while (!this.peek('}')) body.push(this.statement()); this.expect('}'); return { k: 'Block', body }; },
  varDecl() {
    const kind = this.tk[this.i++].v, decls = [];
    do {
      if (this.peek('{') || this.peek('[')) {      // const {mu, sigma} = state;  const [a, b] = state.theta;
        const pat = this.pattern();
        this.expect('=');
        const src = this.assignment();
        if (!isPath(src)) throw 'the right-hand side of a destructuring declaration must be a name or a property/element of one';
        this.bind(pat, src, decls);
        continue;
      }
      const name = this.ident(); let init = null; if (this.eat('=')) init = this.assignment(); decls.push({ name, init });
    } while (this.eat(','));
    return { k: 'VarDecl', kind, decls };
  },
  statement() {
    if (this.peek('{')) return this.block();
    if (this.eat(';')) return { k: 'Empty' };
    if (this.peek('var') || this.peek('let') || this.peek('const')) { const d = this.varDecl(); this.endStmt(); return d; }
    if (this.eat('for')) {
      this.expect('(');
      let init = null, test = null, update = null;
      if ((this.peek('var') || this.peek('let') || this.peek('const')) && (this.peekAt(2, 'of') || this.peekAt(1, '{') || this.peekAt(1, '['))) {
        // for (const x of arr) body   ->   for (var k = 0; k < arr.length; k++) { var x = arr[k]; body }
        const save = this.i, declKind = this.tk[this.i].v;
        this.i++;
        const pat = this.pattern();
        if (this.eat('of')) {
          const arr = this.assignment();
          this.expect(')');
          if (!isPath(arr)) throw 'for-of needs a name or a property/element of one to iterate over';
          const k = this.fresh('k'), decls = [];
          this.bind(pat, { k: 'Index', obj: arr, idx: { k: 'Id', name: k } }, decls);
          const body = this.statement();
          return countedLoop(k, arr, [{ k: 'VarDecl', kind: declKind, decls }].concat(body.k === 'Block' ? body.body : [body]));
        }
        this.i = save;
      }
      if (!this.peek(';')) init = (this.peek('var') || this.peek('let') || this.peek('const')) ? this.varDecl() : { k: 'ExprStmt', expr: this.expression() };
      if (this.peek('in') || this.peek('of')) throw 'for-in loops (and for-of over anything but a declared name) are not supported; use for (var i = 0; i < n; i++)';
      this.expect(';');
      if (!this.peek(';')) test = this.expression();
      this.expect(';');
      if (!this.peek(')')) update = this.expression();
      this.expect(')');
      return { k: 'For', init, test, update, body: this.statement() };
    }
    if (this.eat('while')) { this.expect('('); const test = this.expression(); this.expect(')'); return { k: 'For', init: null, test, update: null, body: this.statement() }; }
    if (this.eat('if')) {
      this.expect('('); const test = this.expression(); this.expect(')');
      const cons = this.statement();
      let alt = null;
      if (this.eat('else')) alt = this.statement();
      return { k: 'If', test, cons, alt };
    }
    if (this.eat('return')) {
      let arg = null;
      const t = this.tk[this.i];
      if (!this.peek(';') && !this.peek('}') && !t.nl && t.t !== 'eof') arg = this.expression();
      this.endStmt();
      return { k: 'Return', arg };
    }
    if (this.eat('break')) { this.endStmt(); return { k: 'Break' }; }
    if (this.eat('continue')) { this.endStmt(); return { k: 'Continue' }; }
    if (this.eat('do')) {          // do body while (c);   ->   for (;;) { body; if (!(c)) break; }
      const body = this.statement();
      this.expect('while'); this.expect('('); const test = this.expression(); this.expect(')'); this.endStmt();
      if (hasOwnJump(body, 'Continue')) throw "'continue' inside a do...while body is not supported";
      const stmts = body.k === 'Block' ? body.body.slice() : [body];
      stmts.push({ k: 'If', test: { k: 'Unary', op: '!', arg: test }, cons: { k: 'Break' }, alt: null });
      return { k: 'For', init: null, test: null, update: null, body: { k: 'Block', body: stmts } };
    }
    if (this.eat('switch')) {      // switch without fall-through -> if / else if chain on a temporary
      this.expect('('); const disc = this.expression(); this.expect(')'); this.expect('{');
      const groups = [];           // {tests: [expr], isDefault, body: [stmt]}
      let cur = null;
      while (!this.peek('}')) {
        if (this.peek('case') || this.peek('default')) {
          if (!cur || cur.body.length) { cur = { tests: [], isDefault: false, body: [] }; groups.push(cur); }
          if (this.eat('default')) cur.isDefault = true; else { this.i++; cur.tests.push(this.expression()); }
          this.expect(':');
        } else {
          if (!cur) throw "expected 'case' inside switch";
          cur.body.push(this.statement());
        }
      }
      this.expect('}');
      const tmp = this.fresh('sw');
      let chain = null, dflt = null;
      groups.forEach((g, gi) => {
        const isLast = gi === groups.length - 1;
        const strip = (list) => {      // removes the case's closing break (also from a trailing block); true if the case cannot fall through
          const q = list[list.length - 1];
          if (!q) return false;
          if (q.k === 'Break') { list.pop(); return true; }
          if (q.k === 'Block') return strip(q.body);
          return q.k === 'Return' || q.k === 'Continue';
        };
        if (!strip(g.body) && !isLast) throw 'a switch case that falls through into the next one is not supported (end it with break)';
        if (g.body.some((st) => hasOwnJump(st, 'Break'))) throw "'break' nested inside a switch case is not supported (only as the last statement of the case)";
      });
      for (let gi = groups.length - 1; gi >= 0; gi--) {
        const g = groups[gi], blk = { k: 'Block', body: g.body };
        if (g.isDefault) { if (gi !== groups.length - 1) throw "'default' must be the last clause of the switch"; dflt = blk; continue; }
        const test = g.tests.map((t) => ({ k: 'Binary', op: '===', l: { k: 'Id', name: tmp }, r: t })).reduce((a, b) => ({ k: 'Logical', op: '||', l: a, r: b }));
        chain = { k: 'If', test, cons: blk, alt: chain || dflt };
      }
      return { k: 'Block', body: [{ k: 'VarDecl', kind: 'var', decls: [{ name: tmp, init: disc }] }].concat(chain ? [chain] : (dflt ? [dflt] : [])) };
    }
    for (const kw of ['throw', 'try', 'function'])
      if (this.peek(kw)) throw "'" + kw + "' statements are not supported inside log_post";
    const expr = this.expression();
    this.endStmt();
    return { k: 'ExprStmt', expr };
  },
  expression() { let e = this.assignment(); while (this.eat(',')) e = { k: 'Seq', l: e, r: this.assignment() }; return e; },
  assignment() {
    const left = this.conditional();
    for (const op of ['=', '+=', '-=', '*=', '/=', '%=']) if (this.peek(op)) { this.i++; return { k: 'Assign', op, target: left, value: this.assignment() }; }
    return left;
  },
  conditional() {
    const test = this.binary(0);
    if (this.eat('?')) { const a = this.assignment(); this.expect(':'); const b = this.assignment(); return { k: 'Cond', test, a, b }; }
    return test;
  },
  binary(level) {
    const LEVELS = [['||'], ['&&'], ['|'], ['^'], ['&'], ['===', '!==', '==', '!='], ['<=', '>=', '<', '>'], ['>>>', '<<', '>>'], ['+', '-'], ['*', '/', '%']];
    if (level === LEVELS.length) return this.unary();
    let left = this.binary(level + 1);
    for (;;) {
      let hit = null;
      for (const op of LEVELS[level]) if (this.peek(op)) { hit = op; break; }
      if (!hit) return left;
      this.i++;
      const right = this.binary(level + 1);
      left = { k: level < 2 ? 'Logical' : 'Binary', op: hit, l: left, r: right };
    }
  },
  unary() {
    for (const op of ['-', '+', '!', '~']) if (this.peek(op)) { this.i++; return { k: 'Unary', op, arg: this.unary() }; }
    for (const op of ['++', '--']) if (this.peek(op)) { this.i++; return { k: 'Update', op, prefix: true, target: this.unary() }; }
    if (this.peek('typeof')) throw "'typeof' is not supported inside log_post";
    if (this.peek('new')) {           // only `new Array(n)`: a local array of n numbers (see NewArray)
      this.i++;
      if (!(this.tk[this.i].t === 'id' && this.tk[this.i].v === 'Array')) throw "'new' is only supported as new Array(n) inside log_post";
    }
    const base = this.postfix();
    if (this.eat('**')) return { k: 'Call', callee: { k: 'Member', obj: { k: 'Id', name: 'Math' }, prop: 'pow' }, args: [base, this.unary()] };   // right-associative
    return base;
  },
  postfix() {
    let e = this.primary();
    for (;;) {
      if (this.eat('.')) { const t = this.tk[this.i]; if (t.t !== 'id') throw "expected a property name after '.'"; this.i++; e = { k: 'Member', obj: e, prop: t.v }; }
      else if (this.eat('[')) {
        const idx = this.expression(); this.expect(']');
        e = idx.k === 'Str' ? { k: 'Member', obj: e, prop: idx.v } : { k: 'Index', obj: e, idx };
      } else if (this.eat('(')) {
        const args = [];
        if (!this.peek(')')) do { args.push(this.assignment()); } while (this.eat(','));
        this.expect(')');
        e = { k: 'Call', callee: e, args };
      } else if ((this.peek('++') || this.peek('--')) && !this.tk[this.i].nl) { e = { k: 'Update', op: this.tk[this.i++].v, prefix: false, target: e }; }
      else return e;
    }
  },
  primary() {
    const t = this.tk[this.i];
    if (t.t === 'num') { this.i++; return { k: 'Num', v: t.v }; }
    if (t.t === 'str') { this.i++; return { k: 'Str', v: t.v }; }
    if (this.peek('(')) {
      // (a, b) => ...   -- try the arrow-parameter reading first, fall back to a parenthesised expression
      const save = this.i;
      this.i++;
      const params = [];
      let ok = true;
      if (!this.peek(')')) {
        do { const q = this.tk[this.i]; if (q.t === 'id' && !KEYWORDS.has(q.v)) { params.push(q.v); this.i++; } else { ok = false; break; } } while (this.eat(','));
      }
      if (ok && this.eat(')') && this.eat('=>')) {
        const body = this.peek('{') ? this.block() : { k: 'Block', body: [{ k: 'Return', arg: this.assignment() }] };
        return { k: 'Func', params, body };
      }
      this.i = save;
      this.i++;
      const e = this.expression(); this.expect(')'); return e;
    }
    if (this.eat('[')) {
      const elems = [];
      if (!this.peek(']')) do { elems.push(this.assignment()); } while (this.eat(','));
      this.expect(']');
      return { k: 'ArrayLit', elems };
    }
    if (t.t === 'id') {
      if (t.v === 'true' || t.v === 'false') { this.i++; return { k: 'Bool', v: t.v === 'true' }; }
      if (t.v === 'function') {          // function expression: only meaningful as `var f = function (...) {...}` (a local helper)
        this.i++;
        if (!this.peek('(')) this.ident();
        this.expect('(');
        const prologue = [];
        const params = this.paramList(prologue);
        this.expect(')');
        return { k: 'Func', params, body: this.functionBody(false, prologue) };
      }
      if (this.peekAt(1, '=>') && !KEYWORDS.has(t.v)) {   // x => ...
        this.i += 2;
        const body = this.peek('{') ? this.block() : { k: 'Block', body: [{ k: 'Return', arg: this.assignment() }] };
        return { k: 'Func', params: [t.v], body };
      }
      if (KEYWORDS.has(t.v)) throw "'" + t.v + "' is not supported inside log_post";
      this.i++;
      return { k: 'Id', name: t.v };
    }
    throw "unexpected '" + t.v + "' in log_post";
  },
};

function hasOwnJump(st, kind) {
  if (!st || typeof st !== 'object') return false;
  if (Array.isArray(st)) return st.some((x) => hasOwnJump(x, kind));
  if (st.k === kind) return true;
  if (st.k === 'For' || st.k === 'Func') return false;
  if (st.k === 'Block') return hasOwnJump(st.body, kind);
  if (st.k === 'If') return hasOwnJump(st.cons, kind) || hasOwnJump(st.alt, kind);
  return false;
}
// `let` / `const` are block-scoped: the same name may be declared again in another block (for (...) { const row = a[i]; } twice) or
// shadow an outer one.  The translator works with function-scoped locals, so every such re-declaration gets a name of its own here.
function uniquifyBlockScoped(fn, P) {
  const seen = new Set(fn.params);
  const declare = (st, env) => {       // let/const declared directly by this statement -> entries of env
    if (!st || st.k !== 'VarDecl') return;
    for (const d of st.decls) {
      if (st.kind === 'var') { seen.add(d.name); continue; }
      if (seen.has(d.name)) env[d.name] = P.fresh('b') + '_' + d.name; else { seen.add(d.name); delete env[d.name]; }
    }
  };
  (function hoistVars(n) {             // `var` declarations are visible in the whole function, wherever they stand
    if (!n || typeof n !== 'object') return;
    if (Array.isArray(n)) { n.forEach(hoistVars); return; }
    if (n.k === 'Func') return;
    if (n.k === 'VarDecl' && n.kind === 'var') n.decls.forEach((d) => seen.add(d.name));
    for (const key of Object.keys(n)) if (key !== 'k') hoistVars(n[key]);
  })(fn.body);
  const go = (node, env) => {
    if (!node || typeof node !== 'object') return node;
    if (Array.isArray(node)) return node.map((x) => go(x, env));
    switch (node.k) {
      case 'Id': return Object.prototype.hasOwnProperty.call(env, node.name) ? { k: 'Id', name: env[node.name] } : node;
      case 'Block': {
        const inner = Object.assign({}, env);
        node.body.forEach((st) => declare(st, inner));
        return { k: 'Block', body: node.body.map((st) => go(st, inner)) };
      }
      case 'For': {
        const inner = Object.assign({}, env);
        declare(node.init, inner);
        return { k: 'For', init: go(node.init, inner), test: go(node.test, inner), update: go(node.update, inner), body: go(node.body, inner) };
      }
      case 'VarDecl': return { k: 'VarDecl', kind: node.kind, decls: node.decls.map((d) => ({ name: (node.kind !== 'var' && Object.prototype.hasOwnProperty.call(env, d.name)) ? env[d.name] : d.name, init: go(d.init, env) })) };
      case 'Func': {
        const inner = Object.assign({}, env);
        node.params.forEach((q) => delete inner[q]);
        return { k: 'Func', params: node.params, body: go(node.body, inner) };
      }
      case 'Member': return { k: 'Member', obj: go(node.obj, env), prop: node.prop };
      default: {
        const o = {};
        for (const key of Object.keys(node)) o[key] = key === 'k' ? node.k : go(node[key], env);
        return o;
      }
    }
  };
  return { params: fn.params, body: go(fn.body, {}) };
}

// `var a = []; for (var g = 0; g < n; g++) a.push(v);` -- an array grown by exactly one push per iteration of a counted loop that starts
// at 0 has length n and element g at index g: rewritten to `var a = Array(n)` and `a[g] = v` (the only form of push that is supported).
function rewritePushLoops(fn) {
  const body = fn.body;
  const empties = new Set();
  walk(body, (x) => { if (x.k === 'VarDecl') x.decls.forEach((d) => { if (d.init && d.init.k === 'ArrayLit' && d.init.elems.length === 0) empties.add(d.name); }); });
  if (!empties.size) return fn;
  const isPush = (st, name) => st && st.k === 'ExprStmt' && st.expr.k === 'Call' && st.expr.callee.k === 'Member' && st.expr.callee.prop === 'push' &&
    st.expr.callee.obj.k === 'Id' && st.expr.callee.obj.name === name && st.expr.args.length === 1;
  const plan = {};        // array name -> {loop, counter, bound}
  for (const name of empties) {
    let pushes = 0, where = null;
    walk(body, (x) => { if (x.k === 'Call' && x.callee.k === 'Member' && x.callee.prop === 'push' && x.callee.obj.k === 'Id' && x.callee.obj.name === name) pushes++; });
    walk(body, (x) => {
      if (x.k !== 'For' || !x.init || !x.test || !x.update) return;
      const stmts = x.body.k === 'Block' ? x.body.body : [x.body];
      if (!stmts.some((st) => isPush(st, name))) return;
      const init = x.init.k === 'VarDecl' ? (x.init.decls.length === 1 ? { name: x.init.decls[0].name, v: x.init.decls[0].init } : null)
        : (x.init.k === 'ExprStmt' && x.init.expr.k === 'Assign' && x.init.expr.op === '=' && x.init.expr.target.k === 'Id' ? { name: x.init.expr.target.name, v: x.init.expr.value } : null);
      const t = x.test, u = x.update;
      const step1 = (u.k === 'Update' && u.op === '++' && u.target.k === 'Id') || (u.k === 'Assign' && u.op === '+=' && u.target.k === 'Id' && u.value.k === 'Num' && u.value.v === 1);
      if (!init || !init.v || init.v.k !== 'Num' || init.v.v !== 0 || !step1 || u.target.name !== init.name) return;
      if (t.k !== 'Binary' || t.op !== '<' || t.l.k !== 'Id' || t.l.name !== init.name) return;
      if (stmts.filter((st) => isPush(st, name)).length !== 1 || containsKind(x.body, 'Break') || containsKind(x.body, 'Continue') || containsKind(x.body, 'Return')) return;
      if (assignedNames(x.body).has(init.name)) return;
      where = { loop: x, counter: init.name, bound: t.r };
    });
    if (where && pushes === 1) plan[name] = where;
  }
  const names = Object.keys(plan);
  if (!names.length) return fn;
  const go = (node) => {
    if (!node || typeof node !== 'object') return node;
    if (Array.isArray(node)) return node.map(go);
    if (node.k === 'VarDecl') return { k: 'VarDecl', kind: node.kind, decls: node.decls.map((d) => (plan[d.name] && d.init && d.init.k === 'ArrayLit' && d.init.elems.length === 0)
      ? { name: d.name, init: { k: 'NewArray', len: plan[d.name].bound, fill: null } } : { name: d.name, init: go(d.init) }) };
    for (const nm of names) if (isPush(node, nm)) return { k: 'ExprStmt', expr: { k: 'Assign', op: '=', target: { k: 'Index', obj: { k: 'Id', name: nm }, idx: { k: 'Id', name: plan[nm].counter } }, value: go(node.expr.args[0]) } };
    const o = {};
    for (const key of Object.keys(node)) o[key] = key === 'k' ? node.k : go(node[key]);
    return o;
  };
  return { params: fn.params, body: go(body) };
}

function parseFunctionSource(src) { const P = new Parser(tokenize(src)); return uniquifyBlockScoped(rewritePushLoops(P.parseFunction()), P); }

// ---- modern-JavaScript sugar, rewritten into the core subset before translation ---------------------------------------
// a side-effect free path: name, name.prop, name[i] ...
function isPath(e) {
  if (e.k === 'Id') return true;
  if (e.k === 'Member') return isPath(e.obj);
  if (e.k === 'Index') return isPath(e.obj) && (e.idx.k === 'Num' || e.idx.k === 'Id' || isPath(e.idx));
  return false;
}
function countedLoop(k, arr, body, start) {
  return { k: 'For', init: { k: 'VarDecl', kind: 'var', decls: [{ name: k, init: { k: 'Num', v: start || 0 } }] },
    test: { k: 'Binary', op: '<', l: { k: 'Id', name: k }, r: { k: 'Member', obj: arr, prop: 'length' } },
    update: { k: 'Update', op: '++', prefix: false, target: { k: 'Id', name: k } }, body: { k: 'Block', body } };
}
function cloneRenamed(node, map) {      // deep copy with the names in `map` replaced (a nested function that re-declares one shadows it)
  if (!node || typeof node !== 'object') return node;
  if (Array.isArray(node)) return node.map((x) => cloneRenamed(x, map));
  if (node.k === 'Id') {
    if (!Object.prototype.hasOwnProperty.call(map, node.name)) return { k: 'Id', name: node.name };
    const to = map[node.name];
    return typeof to === 'string' ? { k: 'Id', name: to } : cloneRenamed(to, {});      // a path substituted for a parameter
  }
  if (node.k === 'Func') {
    const inner = Object.assign({}, map);
    for (const q of node.params) delete inner[q];
    return { k: 'Func', params: node.params.slice(), body: cloneRenamed(node.body, inner) };
  }
  const out = {};
  for (const key of Object.keys(node)) {
    if (key === 'decls') out.decls = node.decls.map((d) => ({ name: (Object.prototype.hasOwnProperty.call(map, d.name) && typeof map[d.name] === 'string') ? map[d.name] : d.name, init: cloneRenamed(d.init, map) }));
    else out[key] = cloneRenamed(node[key], map);
  }
  return out;
}
function declaredIn(node, out) {        // var/let/const names of a function body (not of nested functions)
  if (!node || typeof node !== 'object') return out;
  if (Array.isArray(node)) { node.forEach((x) => declaredIn(x, out)); return out; }
  if (node.k === 'Func') return out;
  if (node.k === 'VarDecl') node.decls.forEach((d) => out.add(d.name));
  for (const key of Object.keys(node)) if (key !== 'k') declaredIn(node[key], out);
  return out;
}
// the body of a callback as statements of the enclosing function: parameters and locals renamed apart, `return` rewritten by `onReturn`
function inlineCallback(fn, argNames, P, onReturn, what) {
  if (!fn || fn.k !== 'Func') throw what + ' needs a function expression or an arrow function as its argument';
  const map = {}, tag = P.fresh('cb') + '_';
  fn.params.forEach((q, i) => { map[q] = i < argNames.length ? argNames[i] : tag + q; });
  if (fn.params.length > argNames.length) throw what + ': the callback takes more parameters than ' + what + ' passes';
  for (const nm of declaredIn(fn.body, new Set())) if (!Object.prototype.hasOwnProperty.call(map, nm)) map[nm] = tag + nm;
  const body = cloneRenamed(fn.body, map);
  const fix = (st, depth) => {
    if (!st || typeof st !== 'object') return st;
    if (Array.isArray(st)) { const o = []; st.forEach((x) => { const r = fix(x, depth); if (Array.isArray(r)) o.push(...r); else o.push(r); }); return o; }
    if (st.k === 'Return') return onReturn(st.arg, depth);
    if (st.k === 'Block') return { k: 'Block', body: fix(st.body, depth) };
    if (st.k === 'If') { const w = (x) => { const r = fix(x, depth); return Array.isArray(r) ? { k: 'Block', body: r } : r; }; return { k: 'If', test: st.test, cons: w(st.cons), alt: st.alt ? w(st.alt) : null }; }
    if (st.k === 'For') { const r = fix(st.body, depth + 1); return Object.assign({}, st, { body: Array.isArray(r) ? { k: 'Block', body: r } : r }); }
    return st;
  };
  return fix(body.body, 0);
}
function isMethodCall(e, name) { return e && e.k === 'Call' && e.callee.k === 'Member' && e.callee.prop === name && isPath(e.callee.obj); }
function isMapCall(e) { return e && e.k === 'Call' && e.callee.k === 'Member' && e.callee.prop === 'map' && e.args.length === 1 && e.args[0].k === 'Func'; }

// arr.forEach(cb) as a statement, and arr.reduce(cb, init) anywhere in the expressions of a statement
function desugarBlock(block, P) {
  const out = [];
  for (const st of block.body) desugarStatement(st, P, out);
  return { k: 'Block', body: out };
}
function desugarStatement(st, P, out) {
  const sub = (x) => { if (!x) return x; const o = []; desugarStatement(x, P, o); return o.length === 1 ? o[0] : { k: 'Block', body: o }; };
  if (st.k === 'Block') { out.push(desugarBlock(st, P)); return; }
  if (st.k === 'If') { const test = hoistReduce(st.test, P, out); out.push({ k: 'If', test, cons: sub(st.cons), alt: sub(st.alt) }); return; }
  if (st.k === 'For') {
    for (const part of [st.init, st.test, st.update]) walk(part, (x) => { if (isMethodCall(x, 'reduce') || isMethodCall(x, 'forEach')) throw 'reduce()/forEach() inside the header of a loop is not supported'; });
    out.push(Object.assign({}, st, { body: sub(st.body) }));
    return;
  }
  if (st.k === 'ExprStmt' && st.expr.k === 'Call' && st.expr.callee.k === 'Member' && st.expr.callee.prop === 'forEach' && isMapCall(st.expr.callee.obj)) {
    const id = hoistReduce(st.expr.callee.obj, P, out);
    desugarStatement({ k: 'ExprStmt', expr: { k: 'Call', callee: { k: 'Member', obj: id, prop: 'forEach' }, args: st.expr.args } }, P, out);
    return;
  }
  if (st.k === 'ExprStmt' && isMethodCall(st.expr, 'forEach')) {
    const arr = st.expr.callee.obj, k = P.fresh('k'), x = P.fresh('x');
    if (st.expr.args.length !== 1) throw 'forEach takes one argument here (no thisArg)';
    const cb = st.expr.args[0];
    const names = cb && cb.params ? [cb.params.length > 0 ? P.fresh('cb') + '_' + cb.params[0] : x, k] : [x, k];
    const body = inlineCallback(cb, names, P, (arg, depth) => {
      if (depth > 0) throw 'a return inside a loop inside a forEach callback is not supported';
      return (arg ? [{ k: 'ExprStmt', expr: arg }] : []).concat([{ k: 'Continue' }]);
    }, 'forEach');
    const inner = [];
    desugarStatement({ k: 'Block', body }, P, inner);
    out.push(countedLoop(k, arr, [{ k: 'VarDecl', kind: 'var', decls: [{ name: names[0], init: { k: 'Index', obj: arr, idx: { k: 'Id', name: k } } }] }].concat(inner[0].body)));
    return;
  }
  if (st.k === 'VarDecl') { out.push({ k: 'VarDecl', kind: st.kind, decls: st.decls.map((d) => ({ name: d.name, init: d.init && d.init.k === 'Func' ? desugarFunc(d.init, P) : hoistReduce(d.init, P, out) })) }); return; }
  if (st.k === 'ExprStmt') { out.push({ k: 'ExprStmt', expr: st.expr.k === 'Assign' && st.expr.value.k === 'Func' ? Object.assign({}, st.expr, { value: desugarFunc(st.expr.value, P) }) : hoistReduce(st.expr, P, out) }); return; }
  if (st.k === 'Return') { out.push({ k: 'Return', arg: hoistReduce(st.arg, P, out) }); return; }
  out.push(st);
}
function desugarFunc(fn, P) { return { k: 'Func', params: fn.params, body: desugarBlock(fn.body, P) }; }
// replaces every arr.reduce(cb, init) inside `e` by a fresh variable and emits `var acc = init; for (...) acc = <cb>` in front
function hoistReduce(e, P, out) {
  if (!e || typeof e !== 'object') return e;
  if (Array.isArray(e)) return e.map((x) => hoistReduce(x, P, out));
  if (e.k === 'Func') return e;
  // x.map(f).reduce(g, init) / x.map(f).map(g): materialise the inner map first, then treat the outer call on its name
  if (e.k === 'Call' && e.callee.k === 'Member' && (e.callee.prop === 'reduce' || e.callee.prop === 'map') && isMapCall(e.callee.obj))
    return hoistReduce({ k: 'Call', callee: { k: 'Member', obj: hoistReduce(e.callee.obj, P, out), prop: e.callee.prop }, args: e.args }, P, out);
  // (operands of ?: && || are evaluated conditionally in JavaScript; a loop hoisted out of one runs regardless, which changes nothing: the
  // subset is pure, terminates, and reads outside arrays are NaN rather than faults)
  // f(state), f(state, data), f(state.theta, data.x): a function that is handed objects cannot become a scalar device function; its body
  // is inlined here (parameters bound to the argument paths, numbers to temporaries, locals renamed apart).  P.env is set by the translator.
  if (P.env && e.k === 'Call' && e.callee.k === 'Id' && e.args.some((a) => P.env.isObject(a))) {
    const fn = P.env.funcOf(e.callee.name);
    if (fn) {
      if ((P.inlineDepth || 0) > 8) throw e.callee.name + '() is inlined more than 8 levels deep (recursion is not supported)';
      if (fn.params.length < e.args.length) throw e.callee.name + '() takes ' + fn.params.length + ' argument(s)';
      const map = {}, tag = P.fresh('fn') + '_', pre = [];
      const written = assignedNames(fn.body);
      fn.params.forEach((q, i) => {
        if (i >= e.args.length) { map[q] = tag + q; pre.push({ k: 'VarDecl', kind: 'var', decls: [{ name: tag + q, init: { k: 'Id', name: 'NaN' } }] }); return; }     // missing argument: undefined
        const arg = hoistReduce(e.args[i], P, out);
        if (P.env.isObject(arg)) { if (written.has(q)) throw e.callee.name + '() assigns to its parameter ' + q + ', which is bound to an object here'; map[q] = arg; }
        else { map[q] = tag + q; pre.push({ k: 'VarDecl', kind: 'var', decls: [{ name: tag + q, init: arg }] }); }
      });
      for (const nm of declaredIn(fn.body, new Set())) if (!Object.prototype.hasOwnProperty.call(map, nm)) map[nm] = tag + nm;
      const stmts = cloneRenamed(fn.body, map).body;
      const lastSt = stmts[stmts.length - 1];
      let returns = 0;
      (function count(n) { if (!n || typeof n !== 'object') return; if (Array.isArray(n)) { n.forEach(count); return; } if (n.k === 'Func') return; if (n.k === 'Return') returns++; for (const key of Object.keys(n)) if (key !== 'k') count(n[key]); })(stmts);
      if (!lastSt || lastSt.k !== 'Return' || !lastSt.arg || returns !== 1)
        throw e.callee.name + '() is handed an object (the state, the data, or an array of them) and has to be inlined, which needs a single return at its end';
      P.inlineDepth = (P.inlineDepth || 0) + 1;
      for (const st of pre.concat(stmts.slice(0, -1))) desugarStatement(st, P, out);
      const result = hoistReduce(lastSt.arg, P, out);
      P.inlineDepth--;
      return result;
    }
  }
  // Array(n), new Array(n), Array(n).fill(v): a local array of n numbers (n a translation-time constant)
  if (e.k === 'Call' && e.callee.k === 'Id' && e.callee.name === 'Array' && e.args.length === 1) return { k: 'NewArray', len: hoistReduce(e.args[0], P, out), fill: null };
  if (e.k === 'Call' && e.callee.k === 'Member' && e.callee.prop === 'fill' && e.args.length === 1) {
    const base = hoistReduce(e.callee.obj, P, out);
    if (base.k === 'NewArray') return { k: 'NewArray', len: base.len, fill: hoistReduce(e.args[0], P, out) };
    throw 'fill() is only supported directly on Array(n)';
  }
  // arr.map(cb): a new local array filled by a loop (the callback inlined; parameters and locals renamed apart)
  if (e.k === 'Call' && e.callee.k === 'Member' && e.callee.prop === 'map' && e.args.length === 1 && e.args[0].k === 'Func') {
    let arr = hoistReduce(e.callee.obj, P, out);
    if (!isPath(arr)) throw 'map() needs a name or a property/element of one to iterate over';
    const k = P.fresh('k'), x = P.fresh('x'), z = P.fresh('map');
    const body = inlineCallback(e.args[0], [x, k], P, (arg, depth) => {
      if (!arg) throw 'the map callback must return a value';
      if (depth > 0) throw 'a return inside a loop inside a map callback is not supported';
      return [{ k: 'ExprStmt', expr: { k: 'Assign', op: '=', target: { k: 'Index', obj: { k: 'Id', name: z }, idx: { k: 'Id', name: k } }, value: arg } }, { k: 'Continue' }];
    }, 'map');
    if (body.length && body[body.length - 1].k === 'Continue') body.pop();
    const inner = [];
    desugarStatement({ k: 'Block', body }, P, inner);
    out.push({ k: 'VarDecl', kind: 'var', decls: [{ name: z, init: { k: 'NewArray', len: { k: 'Member', obj: arr, prop: 'length' }, fill: { k: 'Num', v: 0 } } }] });
    out.push(countedLoop(k, arr, [{ k: 'VarDecl', kind: 'var', decls: [{ name: x, init: { k: 'Index', obj: arr, idx: { k: 'Id', name: k } } }] }].concat(inner[0].body)));
    return { k: 'Id', name: z };
  }
  // arr.some(cb) / arr.every(cb): a 0/1 flag set by a loop that stops at the first decisive element
  if ((isMethodCall(e, 'some') || isMethodCall(e, 'every')) && e.args.length === 1 && e.args[0].k === 'Func') {
    const some = e.callee.prop === 'some', arr = e.callee.obj, k = P.fresh('k'), x = P.fresh('x'), flag = P.fresh(some ? 'some' : 'every');
    const set = { k: 'Block', body: [{ k: 'ExprStmt', expr: { k: 'Assign', op: '=', target: { k: 'Id', name: flag }, value: { k: 'Num', v: some ? 1 : 0 } } }, { k: 'Break' }] };
    const body = inlineCallback(e.args[0], [x, k], P, (arg, depth) => {
      if (!arg) throw 'the ' + e.callee.prop + ' callback must return a value';
      if (depth > 0) throw 'a return inside a loop inside a ' + e.callee.prop + ' callback is not supported';
      return [{ k: 'If', test: some ? arg : { k: 'Unary', op: '!', arg }, cons: set, alt: null }, { k: 'Continue' }];
    }, e.callee.prop);
    if (body.length && body[body.length - 1].k === 'Continue') body.pop();
    const inner = [];
    desugarStatement({ k: 'Block', body }, P, inner);
    out.push({ k: 'VarDecl', kind: 'var', decls: [{ name: flag, init: { k: 'Num', v: some ? 0 : 1 } }] });
    out.push(countedLoop(k, arr, [{ k: 'VarDecl', kind: 'var', decls: [{ name: x, init: { k: 'Index', obj: arr, idx: { k: 'Id', name: k } } }] }].concat(inner[0].body)));
    return { k: 'Binary', op: '!==', l: { k: 'Id', name: flag }, r: { k: 'Num', v: 0 } };
  }
  if (isMethodCall(e, 'reduce')) {
    if (e.args.length !== 1 && e.args.length !== 2) throw 'reduce takes a callback and an optional initial value';
    const arr = e.callee.obj, cb = e.args[0], k = P.fresh('k'), acc = P.fresh('acc'), x = P.fresh('x');
    // without an initial value the first element is the start and the loop begins at the second (an empty array throws in JavaScript)
    const noInit = e.args.length === 1;
    const init = noInit ? { k: 'Index', obj: arr, idx: { k: 'Num', v: 0 } } : hoistReduce(e.args[1], P, out);
    const body = inlineCallback(cb, [acc, x, k], P, (arg, depth) => {
      if (!arg) throw 'the reduce callback must return a value';
      if (depth > 0) throw 'a return inside a loop inside a reduce callback is not supported';
      // acc = acc + t  is spelled  acc += t  (the same operation; the form the lane-splitting analysis knows)
      const asg = (arg.k === 'Binary' && arg.op === '+' && arg.l.k === 'Id' && arg.l.name === acc) ? { k: 'Assign', op: '+=', target: { k: 'Id', name: acc }, value: arg.r }
        : { k: 'Assign', op: '=', target: { k: 'Id', name: acc }, value: arg };
      return [{ k: 'ExprStmt', expr: asg }, { k: 'Continue' }];
    }, 'reduce');
    if (body.length && body[body.length - 1].k === 'Continue') body.pop();      // a trailing continue is a no-op
    const inner = [];
    desugarStatement({ k: 'Block', body }, P, inner);
    out.push({ k: 'VarDecl', kind: 'var', decls: [{ name: acc, init }] });
    out.push(countedLoop(k, arr, [{ k: 'VarDecl', kind: 'var', decls: [{ name: x, init: { k: 'Index', obj: arr, idx: { k: 'Id', name: k } } }] }].concat(inner[0].body), noInit ? 1 : 0));
    return { k: 'Id', name: acc };
  }
  const o = {};
  for (const key of Object.keys(e)) o[key] = key === 'k' ? e.k : hoistReduce(e[key], P, out);
  return o;
}

// ------------------------------------------------------------------------------------------
// AST helpers
function walk(node, f) {
  if (!node || typeof node !== 'object') return;
  if (Array.isArray(node)) { node.forEach((x) => walk(x, f)); return; }
  if (node.k) f(node);
  for (const key of Object.keys(node)) if (key !== 'k') walk(node[key], f);
}
function assignedNames(node) {   // every local name written anywhere inside node
  const s = new Set();
  walk(node, (x) => {
    if (x.k === 'VarDecl') x.decls.forEach((d) => s.add(d.name));
    if ((x.k === 'Assign' || x.k === 'Update') && x.target.k === 'Id') s.add(x.target.name);
  });
  return s;
}
// Definite-assignment walk: true iff no variable of `tracked` is read before it was assigned on
// every path through `stmts` (so it carries nothing from one loop iteration to the next).
function definitelyAssigned(stmts, tracked, defined, acc) {
  const reads = (node) => { for (const nm of idsOf(node)) if (tracked.has(nm) && !defined.has(nm)) return false; return true; };
  for (const st of stmts) {
    switch (st.k) {
      case 'Empty': break;
      case 'Block': if (!definitelyAssigned(st.body, tracked, defined, acc)) return false; break;
      case 'VarDecl':
        for (const d of st.decls) { if (d.init) { if (!reads(d.init)) return false; defined.add(d.name); } }
        break;
      case 'ExprStmt': {
        const e = st.expr;
        if (e.k === 'Assign' && e.target.k === 'Id') {
          if (!reads(e.value)) return false;
          if (acc instanceof Set ? acc.has(e.target.name) : e.target.name === acc) break;
          if (e.op === '=') defined.add(e.target.name);
          else if (tracked.has(e.target.name) && !defined.has(e.target.name)) return false;
        } else if (e.k === 'Update' && e.target.k === 'Id') {
          if (tracked.has(e.target.name) && !defined.has(e.target.name)) return false;
        } else if (!reads(e)) return false;
        break;
      }
      case 'If': {
        if (!reads(st.test)) return false;
        const a = new Set(defined), b = new Set(defined);
        if (!definitelyAssigned([st.cons], tracked, a, acc)) return false;
        if (st.alt && !definitelyAssigned([st.alt], tracked, b, acc)) return false;
        if (st.alt) for (const nm of a) if (b.has(nm)) defined.add(nm);
        break;
      }
      case 'For': {
        if (st.init && !definitelyAssigned([st.init], tracked, defined, acc)) return false;
        if (st.test && !reads(st.test)) return false;
        const inner = new Set(defined);
        if (!definitelyAssigned([st.body], tracked, inner, acc)) return false;
        if (st.update && !definitelyAssigned([{ k: 'ExprStmt', expr: st.update }], tracked, inner, acc)) return false;
        break;
      }
      case 'Return': if (st.arg && !reads(st.arg)) return false; break;
      default: return false;
    }
  }
  return true;
}
function idsOf(node) { const s = new Set(); walk(node, (x) => { if (x.k === 'Id') s.add(x.name); }); return s; }
function containsKind(node, kind) { let hit = false; walk(node, (x) => { if (x.k === kind) hit = true; }); return hit; }

module.exports = { tokenize, parseFunctionSource, desugarBlock, walk, assignedNames, definitelyAssigned, idsOf, containsKind, declaredIn, hasOwnJump };
