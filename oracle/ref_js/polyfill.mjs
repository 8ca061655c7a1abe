// Library functions newer than Node 12 that the reference's code calls (test infrastructure, see downlevel.py).
if (!Object.hasOwn) Object.defineProperty(Object, 'hasOwn', { value: (o, k) => Object.prototype.hasOwnProperty.call(o, k), configurable: true, writable: true });
for (const C of [Array, Uint8Array, Uint32Array, BigUint64Array, String]) {
  if (!C.prototype.at)
    Object.defineProperty(C.prototype, 'at', {
      value(i) {
        i = Math.trunc(i) || 0;
        if (i < 0) i += this.length;
        return i < 0 || i >= this.length ? undefined : this[i];
      },
      configurable: true,
      writable: true,
    });
}
if (!Array.prototype.findLast)
  Object.defineProperty(Array.prototype, 'findLast', {
    value(f) {
      for (let i = this.length - 1; i >= 0; i--) if (f(this[i], i, this)) return this[i];
      return undefined;
    },
    configurable: true,
    writable: true,
  });
if (!String.prototype.replaceAll)
  Object.defineProperty(String.prototype, 'replaceAll', {
    value(a, b) {
      return this.split(a).join(b);
    },
    configurable: true,
    writable: true,
  });
if (typeof globalThis.structuredClone !== 'function') globalThis.structuredClone = (x) => JSON.parse(JSON.stringify(x));
