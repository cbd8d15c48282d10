// MenciusNative.scala -- the Mencius half of the reference-side binding (source only, like Native.scala: no JDK /
// scalac in this image).  Drop into jvm/src/main/scala/frankenpaxos/gpu/ next to Native.scala.
//
//   GpuMenciusEngine       ONE libfpx context = the acceptors of every acceptor group of every leader group + the proxy
//                          leader's tallies, for a deployment whose acceptors and proxy leaders run in one process on
//                          the GPU box (one Transport event loop: not thread-safe, like every actor).  Log window, value
//                          garbage collection and thrifty target windows as in GpuPhase2Engine (Native.scala).
//   GpuMenciusProxyLeader  stands where a mencius.ProxyLeader stands (mencius/ProxyLeaderMain.scala): leaders of EVERY
//                          leader group keep sending it Phase2a and Phase2aNoopRange (mencius/Leader.scala:342-345, 455)
//   GpuMenciusAcceptor     stands at ONE acceptor address (mencius/AcceptorMain.scala); every acceptor address gets one,
//                          all over the same engine.  A Mencius Leader sends Phase1a to ACCEPTOR addresses
//                          (mencius/Leader.scala:486-491, resend timer :288-297): without an actor there a new leader
//                          would never finish Phase 1 (mencius/Acceptor.scala:166-200).
//
// The walk of both on wire bytes -- a leader change in one leader group beside an undisturbed one, commands and noop
// ranges, Nacks to LeaderInbound field 7 -- is tests/test_jni_shim.py::test_a_mencius_leader_change_on_wire_bytes_...
// (mock JVM, against the oracle).
//
// The context's rows are leader-group-major in HBM (include/fpx.h, FPX_F_SLOT_MAJOR_ROWS): a run of commands is handed
// over AS THE LEADER GROUPS' BATCHES BACK TO BACK, each in slot order (a bucket per leader group, which is also the order
// in which one leader's messages arrive).
package frankenpaxos.gpu

import frankenpaxos.Actor
import frankenpaxos.Chan
import frankenpaxos.Logger
import frankenpaxos.mencius._
import frankenpaxos.roundsystem.RoundSystem
import scala.collection.mutable

class GpuMenciusEngine[Transport <: frankenpaxos.Transport[Transport]](
    logger: Logger,
    config: Config[Transport],
    numSlots: Int = 1 << 22,
    retainSlots: Int = 1 << 20,
    thrifty: Boolean = true               // mencius/ProxyLeader.scala:236: rand.shuffle(group).take(config.quorumSize)
) {
  config.checkValid()
  val L: Int = config.numLeaderGroups
  val A: Int = config.acceptorAddresses(0).size            // acceptor groups per leader group
  val R: Int = config.acceptorAddresses(0)(0).size         // acceptors per group (2f + 1)
  // a row keeps its leader group and its acceptor group when the log wraps: slot % L and (slot / L) % A are those of
  // slot % numSlots
  logger.check(numSlots % (L * A) == 0)
  private val chunk = math.max(L * A, numSlots / 16 / (L * A) * (L * A))
  logger.check(retainSlots + 2 * chunk <= numSlots)
  // fpx_config as fpx_jni.c reads it: slots, replicas, groups, leader groups, f, quorum kind, grid rows / cols,
  // leaders per group, ballot model (ACCEPTOR: noop ranges act on the acceptor's round), tally ways, replica base /
  // total, device, flags
  private val handle = Native.create(
    Array(numSlots, R, A, L, config.f, /*THRESHOLD*/ 0, 0, 0, config.leaderAddresses(0).size, /*ACCEPTOR*/ 0, 4, 0, 0, 0, 0))
  if (handle < 0) Native.check((-handle).toInt, logger)

  def leaderGroupOf(slot: Int): Int = slot % L                       // mencius/ProxyLeader.scala:169-176
  def ctxGroup(leaderGroup: Int, acceptorGroup: Int): Int = leaderGroup * A + acceptorGroup

  // ---- value ids, the window: as GpuPhase2Engine (an id lives as long as the row it was proposed in)
  private val values = mutable.ArrayBuffer[CommandBatchOrNoop]()
  private val freeIds = mutable.ArrayStack[Int]()
  private val idsOfRow = Array.fill(numSlots)(List.empty[Int])
  private def intern(row: Int, v: CommandBatchOrNoop): Int =
    if (v.value.isNoop) -1
    else {
      val id = if (freeIds.nonEmpty) freeIds.pop() else { values += null; values.size - 1 }
      values(id) = v
      idsOfRow(row) = id :: idsOfRow(row)
      id
    }
  def valueOf(id: Int): CommandBatchOrNoop =
    if (id < 0) CommandBatchOrNoop().withNoop(Noop()) else values(id)

  private var base = 0
  private val chosenInWindow = new java.util.BitSet(numSlots)
  private var chosenPrefix = 0
  private var highestChosen = -1
  private def row(slot: Int): Int = slot % numSlots
  private def slotOfRow(r: Int): Int = base + ((r - row(base)) % numSlots + numSlots) % numSlots
  private def markChosen(slot: Int): Unit = {
    chosenInWindow.set(row(slot))
    highestChosen = math.max(highestChosen, slot)
    while (chosenPrefix < base + numSlots && chosenInWindow.get(row(chosenPrefix))) chosenPrefix += 1
  }
  private def advanceWindow(): Boolean = {
    var moved = false
    while (chosenPrefix - base >= chunk && highestChosen - (base + chunk) >= retainSlots) {
      val r0 = row(base)
      Native.check(Native.recycleSlots(handle, r0, chunk), logger)   // votes dropped, tallies (ranges too) forgotten
      for (r <- r0 until r0 + chunk) {
        idsOfRow(r).foreach(id => { values(id) = null; freeIds.push(id) })
        idsOfRow(r) = Nil
        chosenInWindow.clear(r)
      }
      base += chunk
      moved = true
    }
    moved
  }
  def inWindow(slot: Int): Boolean = slot >= base && slot < base + numSlots

  // ---- thrifty targets: a window of quorumSize neighbouring acceptors of the slot's group, rotating (any quorumSize of
  // them will do, mencius/ProxyLeader.scala:236)
  private var rotor = 0
  private def thriftyMask(masks: Array[Long], at: Int): Unit = {
    val start = rotor % R
    rotor += 1
    for (j <- 0 until config.quorumSize) { val a = (start + j) % R; masks(at + (a >> 6)) |= 1L << (a & 63) }
  }

  case class Result(chosen: Seq[Chosen], chosenRanges: Seq[ChosenNoopRange], nacks: Seq[(Int, Int, Nack)]) // (slot, round, Nack)

  // ---- one run of commands: mencius.ProxyLeader.handlePhase2a + every mencius.Acceptor.handlePhase2a +
  // mencius.ProxyLeader.handlePhase2b (mencius/ProxyLeader.scala:216-253, 305-353, mencius/Acceptor.scala:202-235)
  private val deferred = mutable.Queue[Phase2a]()
  def commands(incoming: Seq[Phase2a]): Result = {
    val chosenOut = mutable.Buffer[Chosen](); val nackOut = mutable.Buffer[(Int, Int, Nack)]()
    var batch: Seq[Phase2a] = deferred.dequeueAll(_ => true) ++ incoming
    while (batch.nonEmpty) {
      val (nowAny, later) = batch.filter(_.slot >= base).partition(_.slot < base + numSlots)
      // the leader groups' batches back to back, each in slot order; one leader's own messages keep their order
      // (sortBy is stable), messages of different leader groups were in flight together anyway
      val now = nowAny.groupBy(p => leaderGroupOf(p.slot)).toSeq.sortBy(_._1).flatMap(_._2.sortBy(_.slot)).toArray
      val n = now.length
      if (n > 0) {
        val slot = now.map(p => row(p.slot)); val round = now.map(_.round)
        val value = now.map(p => intern(row(p.slot), p.commandBatchOrNoop))
        val masks: Array[Long] = if (thrifty && R > config.quorumSize) {
          val m = new Array[Long](4 * n); for (i <- 0 until n) thriftyMask(m, 4 * i); m
        } else null
        val chosen = new Array[Byte](n); val cr = new Array[Int](n); val cv = new Array[Int](n); val nr = new Array[Int](n)
        Native.check(Native.phase2Fused(handle, n, slot, round, value, masks, chosen, cr, cv, nr), logger)
        for (i <- 0 until n) {
          if (chosen(i) != 0) {                              // mencius/ProxyLeader.scala:338-352
            chosenOut += Chosen(slot = now(i).slot, commandBatchOrNoop = valueOf(cv(i)))
            markChosen(now(i).slot)
          }
          if (nr(i) >= 0) nackOut += ((now(i).slot, round(i), Nack(round = nr(i))))   // mencius/Acceptor.scala:208-219
        }
      }
      batch = if (advanceWindow()) later else { deferred ++= later; Seq.empty }
    }
    Result(chosenOut, Seq.empty, nackOut)
  }

  // ---- one run of noop ranges: the *NoopRange handlers (mencius/ProxyLeader.scala:255-303, 355-411,
  // mencius/Acceptor.scala:237-291), one fused launch (up to 4096 of them walk the chain in one workgroup).  A range is
  // cut at the end of the window; what lies beyond waits like a command beyond the window does
  private val deferredRanges = mutable.Queue[Phase2aNoopRange]()
  def ranges(incoming: Seq[Phase2aNoopRange]): Result = {
    val chosenOut = mutable.Buffer[ChosenNoopRange](); val nackOut = mutable.Buffer[(Int, Int, Nack)]()
    var batch: Seq[Phase2aNoopRange] = deferredRanges.dequeueAll(_ => true) ++ incoming
    while (batch.nonEmpty) {
      val end = base + numSlots
      val now = mutable.Buffer[Phase2aNoopRange](); val later = mutable.Buffer[Phase2aNoopRange]()
      for (p <- batch if p.slotEndExclusive > base) {
        val lo = math.max(p.slotStartInclusive, base)          // (below the window: chosen long ago and recycled)
        if (lo >= end) later += p
        else if (p.slotEndExclusive <= end) now += p.copy(slotStartInclusive = lo)
        else { now += p.copy(slotStartInclusive = lo, slotEndExclusive = end); later += p.copy(slotStartInclusive = end) }
      }
      val m = now.size
      if (m > 0) {
        // rows: a range inside the window is one run of rows of its leader group unless the window wraps inside it --
        // then it is handed over as two ranges
        val parts = now.flatMap { p =>
          val (a, b) = (row(p.slotStartInclusive), row(p.slotEndExclusive - 1) + 1)
          if (a < b) Seq((p, a, b)) else Seq((p, a, numSlots), (p, leaderGroupOf(p.slotStartInclusive), b)).filter(x => x._2 < x._3)
        }
        val k = parts.size
        val start = parts.map(_._2).toArray; val stop = parts.map(_._3).toArray; val round = parts.map(_._1.round).toArray
        val isNew = new Array[Byte](k); val chosen = new Array[Byte](k); val nr = new Array[Int](k)
        val votes = new Array[Long](k * A * 4); val nacks = new Array[Long](k * A * 4)
        Native.check(Native.noopRangesFused(handle, k, A, start, stop, round, null, votes, nacks, nr, isNew, chosen), logger)
        for (((p, _, _), i) <- parts.zipWithIndex) {
          if (chosen(i) != 0) {                              // mencius/ProxyLeader.scala:395-407
            chosenOut += ChosenNoopRange(slotStartInclusive = slotOfRow(start(i)), slotEndExclusive = slotOfRow(start(i)) + (stop(i) - start(i)))
            var s = slotOfRow(start(i))
            while (s < slotOfRow(start(i)) + (stop(i) - start(i))) { if (leaderGroupOf(s) == leaderGroupOf(p.slotStartInclusive)) markChosen(s); s += 1 }
          }
          if (nr(i) >= 0) nackOut += ((p.slotStartInclusive, round(i), Nack(round = nr(i))))   // mencius/Acceptor.scala:245-256
        }
      }
      batch = if (advanceWindow()) later else { deferredRanges ++= later; Seq.empty }
    }
    Result(Seq.empty, chosenOut, nackOut)
  }

  // ---- Phase 1, acceptor side (mencius/Acceptor.scala:166-200)
  def handlePhase1a(leaderGroup: Int, acceptorGroup: Int, index: Int, phase1a: Phase1a): Either[Nack, Phase1b] = {
    val g = ctxGroup(leaderGroup, acceptorGroup)
    val target = new Array[Long](4); target(index >> 6) = 1L << (index & 63)
    val bits = new Array[Long](8)
    // rows, not slots (see GpuPhase2Engine.handlePhase1a): promise from row 0 on, filter the info by slot below
    Native.check(Native.acceptorPhase1a(handle, g, phase1a.round, 0, target, bits), logger)
    if ((bits(4 + (index >> 6)) & (1L << (index & 63))) != 0)
      return Left(Nack(round = Native.acceptorRound(handle, g, index)))               // :173-180
    var cap = 1024
    var slots = new Array[Int](cap); var vr = new Array[Int](cap); var vv = new Array[Int](cap)
    var k = Native.acceptorPhase1bInfo(handle, g, index, 0, cap, slots, vr, vv)
    if (k > cap) {
      cap = k; slots = new Array[Int](cap); vr = new Array[Int](cap); vv = new Array[Int](cap)
      k = Native.acceptorPhase1bInfo(handle, g, index, 0, cap, slots, vr, vv)
    }
    if (k < 0) Native.check(-k, logger)
    val info = (0 until k)                                                             // :184-199
      .map(j => Phase1bSlotInfo(slot = slotOfRow(slots(j)), voteRound = vr(j), voteValue = valueOf(vv(j))))
      .filter(_.slot >= phase1a.chosenWatermark)
      .sortBy(_.slot)
    Right(Phase1b(groupIndex = acceptorGroup, acceptorIndex = index, round = phase1a.round, info = info))
  }

  def close(): Unit = Native.check(Native.destroy(handle), logger)
}

class GpuMenciusProxyLeader[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    engine: GpuMenciusEngine[Transport]
) extends Actor(address, transport, logger) {
  override type InboundMessage = ProxyLeaderInbound
  override val serializer = ProxyLeaderInboundSerializer

  private val slotSystem = new RoundSystem.ClassicRoundRobin(config.numLeaderGroups)
  private val roundSystem = new RoundSystem.ClassicRoundRobin(config.leaderAddresses(0).size)
  private val leaders = for (group <- config.leaderAddresses)
    yield for (a <- group) yield chan[Leader[Transport]](a, Leader.serializer)
  private val replicas = for (a <- config.replicaAddresses) yield chan[Replica[Transport]](a, Replica.serializer)

  // the burst, in arrival order: Left = a command, Right = a noop range
  private val pending = mutable.Buffer[Either[Phase2a, Phase2aNoopRange]]()
  private val tick = timer("gpuMenciusTick", java.time.Duration.ZERO, () => flushTick())

  override def receive(src: Transport#Address, inbound: ProxyLeaderInbound): Unit = {
    import ProxyLeaderInbound.Request
    inbound.request match {
      case Request.Phase2A(p) =>
        if (pending.isEmpty) tick.start()
        pending += Left(p)
      case Request.Phase2ANoopRange(p) =>
        if (pending.isEmpty) tick.start()
        pending += Right(p)
      case Request.HighWatermark(h) =>                       // mencius/ProxyLeader.scala:207-214
        for (group <- leaders; leader <- group) leader.send(LeaderInbound().withHighWatermark(h))
      case Request.Phase2B(_) | Request.Phase2BNoopRange(_) =>
        logger.fatal("GpuMenciusProxyLeader tallies on the device; it never receives Phase2b messages.")
      case Request.Empty =>
        logger.fatal("Empty ProxyLeaderInbound encountered.")
    }
  }

  private def deliver(r: engine.Result): Unit = {
    for (c <- r.chosen) replicas.foreach(_.send(ReplicaInbound().withChosen(c)))
    for (c <- r.chosenRanges) replicas.foreach(_.send(ReplicaInbound().withChosenNoopRange(c)))
    for ((slot, round, nack) <- r.nacks)                     // mencius/Acceptor.scala:215-217
      leaders(slotSystem.leader(slot))(roundSystem.leader(round)).send(LeaderInbound().withNack(nack))
  }

  // The burst is cut into MAXIMAL RUNS of one kind, in arrival order: one native call per run.  A leader's own stream --
  // a command in round r, a noop range in round r' >= r, a command in r' -- reaches the acceptors in the order it was sent
  // (an earlier version flushed all commands, then all ranges: the range of a later round would have made the acceptors
  // Nack the same leader's earlier command).  Within a run of commands the leader groups' batches are regrouped (above).
  private def flushTick(): Unit = {
    var i = 0
    while (i < pending.size) {
      var j = i
      while (j < pending.size && pending(j).isLeft == pending(i).isLeft) j += 1
      val run = pending.slice(i, j)
      deliver(if (pending(i).isLeft) engine.commands(run.map(_.left.get)) else engine.ranges(run.map(_.right.get)))
      i = j
    }
    pending.clear()
  }
}

class GpuMenciusAcceptor[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    engine: GpuMenciusEngine[Transport]
) extends Actor(address, transport, logger) {
  override type InboundMessage = AcceptorInbound
  override val serializer = AcceptorInboundSerializer

  // config.acceptorAddresses(leaderGroup)(acceptorGroup)(index)
  private val (leaderGroup, acceptorGroup, index) = (for {
    (lg, l) <- config.acceptorAddresses.zipWithIndex
    (ag, a) <- lg.zipWithIndex
    (addr, i) <- ag.zipWithIndex
    if addr == address
  } yield (l, a, i)).head

  override def receive(src: Transport#Address, inbound: AcceptorInbound): Unit = {
    inbound.request match {
      case AcceptorInbound.Request.Phase1A(phase1a) =>
        val leader = chan[Leader[Transport]](src, Leader.serializer)
        engine.handlePhase1a(leaderGroup, acceptorGroup, index, phase1a) match {
          case Left(nack)     => leader.send(LeaderInbound().withNack(nack))        // mencius/Acceptor.scala:173-180
          case Right(phase1b) => leader.send(LeaderInbound().withPhase1B(phase1b))  // :184-199
        }
      case AcceptorInbound.Request.Phase2A(_) | AcceptorInbound.Request.Phase2ANoopRange(_) =>
        // the reference's leaders send these to PROXY LEADERS (mencius/Leader.scala:342-345, 455); a deployment that
        // points them at GpuMenciusProxyLeader never delivers one here
        logger.fatal("GpuMenciusAcceptor: Phase2a / Phase2aNoopRange go to GpuMenciusProxyLeader in this deployment.")
      case AcceptorInbound.Request.Empty =>
        logger.fatal("Empty AcceptorInbound encountered.")
    }
  }
}
