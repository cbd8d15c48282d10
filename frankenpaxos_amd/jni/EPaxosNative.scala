// EPaxosNative.scala -- the EPaxos part of the reference-side binding (source only, like Native.scala: no JDK / scalac in
// this image).  Drop into jvm/src/main/scala/frankenpaxos/gpu/ next to Native.scala.
//
//   GpuEPaxosEngine    ONE libfpx EPaxos context = the conflict indices and command logs of ALL n replicas, for a
//                      deployment whose replicas run in one process on the GPU box (one Transport event loop).
//   GpuEPaxosReplica   stands at ONE replica address (epaxos/ReplicaMain.scala); every replica address gets one, all over
//                      the same engine.  `receive` (epaxos/Replica.scala:1081-1119) only enqueues; one zero-delay
//                      Transport timer per burst flushes the queue, one native call per kind of message:
//                        ClientRequest  -> a pre-accept tick (Native.epxPreaccept: transitionToPreAcceptPhase at the
//                                          leaders, handlePreAccept at the others, handlePreAcceptOk -- K5), then
//                                          Native.epxAccept for the commands that took the slow path (K6), then Commit
//                        PreAccept      -> Native.epxHandlePreaccept (K7: a re-sent PreAccept, a recovering replica's)
//                        Accept         -> Native.epxAccept           Prepare -> Native.epxPrepare
//                      and `send`s what the Scala handlers would have sent.  Between hosted replicas nothing crosses the
//                      transport: the PreAccept / PreAcceptOk / Accept / AcceptOk of a tick are the kernels' own traffic.
//
// The same natives driven on wire bytes -- two conflicting commands met in different orders, differing PreAcceptOks, the
// slow path, Accept, AcceptOk, Commit -- against the oracle: tests/test_jni_shim.py::test_an_epaxos_slow_path_commit_...
//
// Execution (round 5; ADVICE r04): a committed instance goes through the reference's OWN classes -- a
// frankenpaxos.depgraph.DependencyGraph[Instance, Int, InstancePrefixSet] (the one ReplicaMain builds, :127), a
// frankenpaxos.statemachine.StateMachine and a frankenpaxos.clienttable.ClientTable per actor, exactly as
// Replica.commit / execute / executeCommand use them (:859-960): every hosted replica executes every commit, the
// instance's leader answers the client (:951-960), a request whose answer is in the client table is answered from it
// (:1121-1147).  A Commit from a replica outside is recorded in the device's command log and conflict index
// (Native.epxHandleCommit: Replica.handleCommit :1567-1575) and executed like the others.
//
// Scope (DESIGN.md section 8): single-key get / set commands of the key-value store (statemachine/KeyValueStore.scala),
// top-one dependencies, sequence number 0; the leader-side recovery timers (Replica.scala:1021-1078) stay with a
// reference Replica if one is wanted -- Prepare / PrepareOk are answered here, not originated; the answers carry what
// the reference's handlePrepareOk reads (:1819-1843: sequenceNumber and dependencies of a PreAccepted / Accepted entry,
// the Commit for a committed one, Replica.nullBallot for an instance never seen).  `leaderStates` (the
// instances a replica is leading, :1243-1249 "stop leading when a larger ballot arrives") is empty between two bursts
// here by construction: an instance this actor leads is pre-accepted, accepted if need be and committed inside the
// flush that proposed it (flushTick checks the f + 1 votes), so a PreAccept in a larger ballot for it finds a
// CommittedEntry and is answered with the Commit (:1228-1238, K7) -- there is no leader state or timer left to drop.
package frankenpaxos.gpu

import com.google.protobuf.ByteString
import frankenpaxos.Actor
import frankenpaxos.Logger
import frankenpaxos.clienttable.ClientTable
import frankenpaxos.depgraph.DependencyGraph
import frankenpaxos.epaxos._
import frankenpaxos.statemachine.{GetRequest, KeyValueStoreInput, SetRequest, StateMachine}
import scala.collection.mutable

class GpuEPaxosEngine[Transport <: frankenpaxos.Transport[Transport]](
    logger: Logger,
    config: Config[Transport],
    numKeys: Int = 1 << 10,
    numInstances: Int = 1 << 20          // instances (leader, number < numInstances) the command logs hold
) {
  val n: Int = config.n
  val handle: Long = Native.epxCreateWithLog(n, numKeys, 0, numInstances)
  if (handle < 0) Native.check((-handle).toInt, logger)

  // a command's triple id is its index here (the GPU carries the int32; Accept and Commit name a triple by it)
  val triples = mutable.ArrayBuffer[CommandOrNoop]()
  // instance -> the dependencies it committed with: n watermarks + the end of the own column's explicit ids (0 = none)
  val depsOf = mutable.Map[(Int, Int), (Array[Int], Int)]()
  // triple id -> the dependencies an Accept carried (the device names an accepted triple by its id alone)
  val tripleDeps = mutable.Map[Int, (Array[Int], Int)]()
  // the actors over this engine: what one of them commits inside its flush the others learn here, not over the transport
  val actors = mutable.Buffer[GpuEPaxosReplica[Transport]]()
  val nextNumber: Array[Int] = Array.fill(n)(0)             // Replica.nextAvailableInstance, per hosted leader
  private val keyIds = mutable.Map[String, Int]()
  def keyOf(k: String): Int = keyIds.getOrElseUpdate(k, { logger.check(keyIds.size < numKeys); keyIds.size })

  // (key id, is set) of a key-value-store command; anything else is outside the device's conflict model
  def classify(c: Command): (Int, Boolean) =
    KeyValueStoreInput.parseFrom(c.command.toByteArray).request match {
      case KeyValueStoreInput.Request.GetRequest(GetRequest(Seq(k)))             => (keyOf(k), false)
      case KeyValueStoreInput.Request.SetRequest(SetRequest(Seq(kv)))            => (keyOf(kv.key), true)
      case _ => logger.fatal("GpuEPaxosEngine: single-key get / set commands only (DESIGN.md section 8).")
    }

  def close(): Unit = Native.check(Native.epxDestroy(handle), logger)
}

class GpuEPaxosReplica[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    engine: GpuEPaxosEngine[Transport],
    // what epaxos/ReplicaMain.scala hands a Replica (:120-135): this replica's copy of the state machine and its graph
    stateMachine: StateMachine,
    dependencyGraph: DependencyGraph[Instance, Int, InstancePrefixSet],
    // true: committed instances are ordered by libfpx's device dependency graph (Native.epxExecute, 2.2 - 2.6e9 commands/s
    // at 2^20 instances per call, profiles/r06_depgraph_dev.md) instead of `dependencyGraph` (the reference's, 2 - 3e7/s);
    // the state machine, the client table and the replies stay where they are.  Default off: the reference's own graph.
    deviceExecution: Boolean = false
) extends Actor(address, transport, logger) {
  override type InboundMessage = ReplicaInbound
  override val serializer = ReplicaInboundSerializer

  private val n = config.n
  private val index = config.replicaAddresses.indexOf(address)
  logger.check(index >= 0)
  private val replicas = for (a <- config.replicaAddresses) yield chan[Replica[Transport]](a, Replica.serializer)
  engine.actors += this

  // ---- execution: Replica.commit's tail, execute and executeCommand (epaxos/Replica.scala:859-960) on the reference's classes
  implicit private val addressSerializer = transport.addressSerializer
  private val clientTable = ClientTable[(Transport#Address, Int), Array[Byte]]()
  private val committed = mutable.Map[Instance, CommandOrNoop]()       // CommittedEntry.triple.commandOrNoop, until executed
  private val executables = mutable.Buffer[Instance]()
  private val blockers = mutable.Set[Instance]()

  // every actor of the process learns every commit: from its own flush, from another hosted actor's, or from a Commit message
  // ---- deviceExecution: what this replica knows of every leader's column above its executed prefix.  The device graph
  // wants DENSE columns (include/fpx.h, fpx_epx_execute): per leader the instances first .. first + count - 1, each once,
  // everything below `first` executed.  So: executedPrefix(l) = the first number of leader l not executed here yet;
  // pendingDeps = the committed instances at or above it that are not executed (watermarks, end of the own column's
  // explicit ids); executedAbove = the ones above it that are (executed out of column order: handed in again as
  // committed -- the device puts them in the order, the actor skips them); a number in the range with no Commit yet is
  // handed in as "not committed": it and whatever reaches it wait (Replica.scala:883-917 `blockers`).
  private val executedPrefix = Array.fill(n)(0)
  private val pendingDeps = mutable.Map[Instance, (Array[Int], Int)]()
  private val executedAbove = mutable.Map[Instance, (Array[Int], Int)]()
  private def ownEnd(instance: Instance, deps: InstancePrefixSetProto): Int = {
    val v = deps.intPrefixSet(instance.replicaIndex).value
    if (v.isEmpty) 0 else v.max + 1
  }

  def learnCommit(instance: Instance, commandOrNoop: CommandOrNoop, dependencies: InstancePrefixSetProto): Unit = {
    if (committed.contains(instance)) return                            // a re-sent Commit
    if (deviceExecution) {
      if (instance.instanceNumber < executedPrefix(instance.replicaIndex) || executedAbove.contains(instance)) return
      committed(instance) = commandOrNoop
      pendingDeps(instance) = (watermarks(dependencies), ownEnd(instance, dependencies))
      return
    }
    committed(instance) = commandOrNoop
    dependencyGraph.commit(instance, 0, InstancePrefixSet.fromProto(dependencies))       // :866-868
  }
  private def executeGraphOnDevice(): Unit = {
    if (pendingDeps.isEmpty) return
    val first = executedPrefix.clone()
    val count = Array.tabulate(n)(l => {
      val known = (pendingDeps.keysIterator ++ executedAbove.keysIterator).filter(_.replicaIndex == l).map(_.instanceNumber)
      if (known.isEmpty) 0 else known.max + 1 - first(l)
    })
    val m = count.sum
    val leader = new Array[Int](m); val number = new Array[Int](m); val deps = new Array[Int](m * n); val ends = new Array[Int](m)
    val isCommitted = new Array[Byte](m)
    var i = 0
    for (l <- 0 until n; x <- first(l) until first(l) + count(l)) {
      leader(i) = l; number(i) = x
      pendingDeps.get(Instance(l, x)).orElse(executedAbove.get(Instance(l, x))) match {
        case Some((w, end)) => Array.copy(w, 0, deps, i * n, n); ends(i) = end; isCommitted(i) = 1
        case None           => ()                                        // no Commit yet: blocks what reaches it
      }
      i += 1
    }
    val order = new Array[Int](m); val component = new Array[Int](m); val counts = new Array[Int](3)
    Native.check(Native.epxExecute(engine.handle, m, n, leader, number, deps, ends, isCommitted, first, count, order, component, counts), logger)
    if (counts(2) != 0) {
      // two different closures with one sum and one 22-bit hash (probability ~2^-22 per pair: include/fpx.h): this batch
      // goes through the reference's graph, which is told what has been executed so far (DependencyGraph.updateExecuted)
      logger.warn("GpuEPaxosReplica: the device dependency graph asks for the host path; this batch runs on the reference's graph.")
      // what this replica has executed: the prefixes, and the instances above them (depgraph/DependencyGraph.scala:176-186)
      val done = InstancePrefixSet.fromWatermarks(executedPrefix.toBuffer)
      executedAbove.keysIterator.foreach(done.add)
      dependencyGraph.updateExecuted(done)
      for ((inst, (w, end)) <- pendingDeps)
        dependencyGraph.commit(inst, 0, InstancePrefixSet.fromProto(prefixSet(w, inst.replicaIndex, end, inst.instanceNumber)))
      dependencyGraph.appendExecute(None, executables, blockers)
    } else {
      for (p <- 0 until counts(0)) executables += Instance(leader(order(p)), number(order(p)))
    }
    for (inst <- executables if !executedAbove.contains(inst)) {
      committed.remove(inst) match {
        case None                => logger.fatal(s"Instance $inst is ready for execution but was never committed here.")
        case Some(commandOrNoop) => executeCommand(inst, commandOrNoop)
      }
      executedAbove(inst) = pendingDeps.remove(inst).get
    }
    executables.clear(); blockers.clear()
    for (l <- 0 until n) {                                               // the executed prefixes move up; what lies below them is forgotten
      while (executedAbove.remove(Instance(l, executedPrefix(l))).isDefined) executedPrefix(l) += 1
    }
  }
  def executeGraph(): Unit = {                                           // :883-917
    if (deviceExecution) { executeGraphOnDevice(); return }
    dependencyGraph.appendExecute(None, executables, blockers)
    for (i <- executables) {
      committed.remove(i) match {
        case None                => logger.fatal(s"Instance $i is ready for execution but was never committed here.")
        case Some(commandOrNoop) => executeCommand(i, commandOrNoop)
      }
    }
    executables.clear(); blockers.clear()
  }
  private def executeCommand(instance: Instance, commandOrNoop: CommandOrNoop): Unit = commandOrNoop.value match {   // :919-965
    case CommandOrNoop.Value.Empty   => logger.fatal("Empty CommandOrNoop.")
    case CommandOrNoop.Value.Noop(_) => ()
    case CommandOrNoop.Value.Command(Command(clientAddressBytes, clientPseudonym, clientId, command)) =>
      val clientAddress = transport.addressSerializer.fromBytes(clientAddressBytes.toByteArray)
      val clientIdentity = (clientAddress, clientPseudonym)
      clientTable.executed(clientIdentity, clientId) match {
        case ClientTable.Executed(_) => ()                               // never execute a command twice
        case ClientTable.NotExecuted =>
          val output = stateMachine.run(command.toByteArray)
          clientTable.execute(clientIdentity, clientId, output)
          if (index == instance.replicaIndex)                            // the instance's leader answers the client
            chan[Client[Transport]](clientAddress, Client.serializer).send(
              ClientInbound().withClientReply(ClientReply(clientPseudonym = clientPseudonym, clientId = clientId,
                                                          result = ByteString.copyFrom(output))))
      }
  }

  private val requests = mutable.Buffer[(Transport#Address, ClientRequest)]()
  private val preAccepts = mutable.Buffer[(Transport#Address, PreAccept)]()
  private val accepts = mutable.Buffer[(Transport#Address, Accept)]()
  private val prepares = mutable.Buffer[(Transport#Address, Prepare)]()
  private val commits = mutable.Buffer[Commit]()
  private var queued = 0
  private val tick = timer("gpuEPaxosTick", java.time.Duration.ZERO, () => flushTick())
  private def enqueue[T](q: mutable.Buffer[T], x: T): Unit = { if (queued == 0) tick.start(); q += x; queued += 1 }

  override def receive(src: Transport#Address, inbound: ReplicaInbound): Unit = {
    import ReplicaInbound.Request
    inbound.request match {
      case Request.ClientRequest(r) =>
        // Replica.handleClientRequest (:1121-1147): an answer that is already in the client table is relayed, a stale
        // request ignored; everything else is led in the next flush
        clientTable.executed((src, r.command.clientPseudonym), r.command.clientId) match {
          case ClientTable.NotExecuted    => enqueue(requests, (src, r))
          case ClientTable.Executed(None) => ()
          case ClientTable.Executed(Some(output)) =>
            chan[Client[Transport]](src, Client.serializer).send(
              ClientInbound().withClientReply(ClientReply(clientPseudonym = r.command.clientPseudonym,
                                                          clientId = r.command.clientId, result = ByteString.copyFrom(output))))
        }
      case Request.PreAccept(r)     => enqueue(preAccepts, (src, r))
      case Request.Accept(r)        => enqueue(accepts, (src, r))
      case Request.Prepare(r)       => enqueue(prepares, (src, r))
      case Request.Commit(r)        => enqueue(commits, r)   // from a replica outside this process (Replica.handleCommit :1567)
      case Request.PreAcceptOk(_) | Request.AcceptOk(_) | Request.PrepareOk(_) | Request.Nack(_) =>
        // replies to a leader role: between hosted replicas they never leave the device; from a replica outside, they
        // belong to a reference Replica that originated the round (see the scope note above)
        logger.debug("GpuEPaxosReplica: a reply addressed to a leader role this actor does not play.")
      case Request.Empty => logger.fatal("Empty ReplicaInbound encountered.")
    }
  }

  private def bit(r: Int): Byte = (1 << r).toByte
  private def watermarks(deps: InstancePrefixSetProto): Array[Int] = deps.intPrefixSet.map(_.watermark).toArray
  private def prefixSet(w: Array[Int], own: Int, valuesEnd: Int, number: Int): InstancePrefixSetProto =
    InstancePrefixSetProto(numReplicas = n, intPrefixSet = w.indices.map(l =>
      // the own-leader column carries the explicit ids number + 1 .. valuesEnd - 1 (dependencies.subtractOne, :582)
      frankenpaxos.compact.IntPrefixSetProto(watermark = w(l), value = if (l == own && valuesEnd > 0) (number + 1 until valuesEnd) else Seq())))

  private def flushTick(): Unit = {
    // ---- the burst's client requests, led by THIS replica: one pre-accept tick (epaxos/Replica.scala:1121-1157,
    // 633-729, 1159-1419).  The tick's delivery order at every replica is the arrival order here; the PreAccept goes to
    // all other replicas (ThriftySystem.NotThrifty, :83), the first n - 2 answers are counted (:1376)
    val m = requests.size
    if (m > 0) {
      val leader = Array.fill(m)(index); val number = new Array[Int](m)
      val key = new Array[Int](m); val isSet = new Array[Byte](m)
      val resp = new Array[Byte](m); val seen = new Array[Byte](m)
      val rank = Array.tabulate(n * m)(i => i % m)
      val others = (0 until n).filter(_ != index)
      val counted = others.take(n - 2).map(1 << _).sum.toByte; val all = others.map(1 << _).sum.toByte
      val first = engine.triples.size
      for (((_, r), i) <- requests.zipWithIndex) {
        number(i) = engine.nextNumber(index); engine.nextNumber(index) += 1
        val (k, set) = engine.classify(r.command); key(i) = k; isSet(i) = if (set) 1 else 0
        resp(i) = counted; seen(i) = all
        engine.triples += CommandOrNoop().withCommand(r.command)
      }
      val fast = new Array[Byte](m); val deps = new Array[Int](m * n); val ldeps = new Array[Int](m * n); val ends = new Array[Int](2 * m)
      Native.check(Native.epxPreaccept(engine.handle, m, n, leader, number, key, isSet, resp, seen, rank, fast, deps, ldeps, ends), logger)
      // the slow path: Accept with the union of the answers (preAcceptingSlowPath :796-813), f other replicas + the proposer
      val slow = (0 until m).filter(fast(_) == 0).toArray
      if (slow.nonEmpty) {
        val k = slow.length
        val tgt = Array.fill(k)(others.take(config.f).map(1 << _).sum.toByte)
        val replies = new Array[Byte](4 * k); val nb = new Array[Int](k)
        Native.check(Native.epxAccept(engine.handle, k, slow.map(leader), slow.map(number), Array.fill(k)(0), Array.fill(k)(index),
                                      slow.map(first + _), slow.map(key), slow.map(isSet), tgt, replies, nb), logger)
        for ((i, j) <- slow.zipWithIndex) logger.check(replies(3 * k + j) != 0)   // f + 1 votes: committed (no competing ballot exists)
      }
      // commit (:815-860): every replica outside this process learns it; the hosted ones already hold the CommittedEntry
      for (i <- 0 until m) {
        val w = deps.slice(i * n, (i + 1) * n)
        engine.depsOf((index, number(i))) = (w, ends(2 * i))
        val commit = Commit(instance = Instance(index, number(i)), commandOrNoop = engine.triples(first + i), sequenceNumber = 0,
                            dependencies = prefixSet(w, index, ends(2 * i), number(i)))
        for ((a, r) <- config.replicaAddresses.zipWithIndex if !hosted(a)) replicas(r).send(ReplicaInbound().withCommit(commit))
        engine.actors.foreach(_.learnCommit(commit.instance, commit.commandOrNoop, commit.dependencies))
      }
      requests.clear()
      engine.actors.foreach(_.executeGraph())   // :869-873 with executeGraphBatchSize = the burst
    }
    // ---- Commits from replicas outside (Replica.handleCommit :1567-1575): the device's command log and conflict index at
    // THIS replica learn them (a later Prepare / PreAccept for the instance is answered with the Commit), then the graph
    if (commits.nonEmpty) {
      // The device's command log holds an instance's own-leader column as (watermark <= number, explicit ids number + 1 ..
      // end - 1): what dependencies.subtractOne(instance) makes of a cover (Replica.scala:582).  A reference Replica may send
      // any InstancePrefixSet; a Commit whose own column is not of that shape is still learnt by the graph below, but the
      // device's log does not take it -- logged and dropped there, not a fatal (ADVICE r05: input from outside is not an
      // invariant of this process).  Of several Commits for one instance in the burst the LAST one wins whole, as when the
      // reference applies them in order (the library takes care of that: fpx_epx_handle_commit).
      def deviceShape(c: Commit): Boolean = {
        val own = c.dependencies.intPrefixSet(c.instance.replicaIndex)
        val x = c.instance.instanceNumber
        own.value.isEmpty || (own.watermark <= x && own.value.sorted == (x + 1 to own.value.max))
      }
      val (forDevice, notForDevice) = commits.partition(deviceShape)
      notForDevice.foreach(c => logger.warn(s"GpuEPaxosReplica: Commit of ${c.instance} carries an own-leader column the " +
                                            "device's command log does not represent; learnt by the graph only."))
      val endsOf = mutable.Map[Commit, Int]()
      if (forDevice.nonEmpty) {
        val k = forDevice.size
        val keyset = forDevice.map(c => if (c.commandOrNoop.value.isNoop) (-1, false) else engine.classify(c.commandOrNoop.getCommand))
        val first = engine.triples.size; forDevice.foreach(c => engine.triples += c.commandOrNoop)
        val own = forDevice.map(c => c.dependencies.intPrefixSet(c.instance.replicaIndex).value)
        val endsIn = own.map(v => if (v.isEmpty) 0 else v.max + 1).toArray
        Native.check(Native.epxHandleCommit(engine.handle, k, n, forDevice.map(_.instance.replicaIndex).toArray,
                                            forDevice.map(_.instance.instanceNumber).toArray, Array.tabulate(k)(first + _),
                                            keyset.map(_._1).toArray, keyset.map(x => (if (x._2) 1 else 0).toByte).toArray,
                                            forDevice.flatMap(c => watermarks(c.dependencies)).toArray, endsIn,
                                            Array.fill(k)(bit(index))), logger)
        for ((c, i) <- forDevice.zipWithIndex) endsOf(c) = endsIn(i)
      }
      for (c <- commits) {
        endsOf.get(c).foreach(end => engine.depsOf((c.instance.replicaIndex, c.instance.instanceNumber)) = (watermarks(c.dependencies), end))
        learnCommit(c.instance, c.commandOrNoop, c.dependencies)
      }
      commits.clear()
      executeGraph()
    }
    // ---- PreAccepts from replicas outside (a leader's re-sent PreAccept, a recovering replica's): Replica.handlePreAccept
    // in full (:1159-1289) at THIS replica; Nack / PreAcceptOk / Commit back to the sender
    if (preAccepts.nonEmpty) {
      val k = preAccepts.size
      val ps = preAccepts.map(_._2)
      val keyset = ps.map(p => if (p.commandOrNoop.value.isNoop) (-1, false) else engine.classify(p.commandOrNoop.getCommand))
      val first = engine.triples.size; ps.foreach(p => engine.triples += p.commandOrNoop)
      val depsIn = ps.flatMap(p => watermarks(p.dependencies)).toArray
      val endsIn = ps.map(p => { val v = p.dependencies.intPrefixSet(p.instance.replicaIndex).value; if (v.isEmpty) 0 else v.max + 1 }).toArray
      val replies = new Array[Byte](4 * k); val nb = new Array[Int](k); val rd = new Array[Int](k * n * n); val ret = new Array[Int](2 * k * n)
      Native.check(Native.epxHandlePreaccept(engine.handle, k, n, ps.map(_.instance.replicaIndex).toArray, ps.map(_.instance.instanceNumber).toArray,
                                             ps.map(_.ballot.ordering).toArray, ps.map(_.ballot.replicaIndex).toArray, keyset.map(_._1).toArray,
                                             keyset.map(x => (if (x._2) 1 else 0).toByte).toArray, Array.tabulate(k)(first + _), depsIn, endsIn,
                                             Array.fill(k)(bit(index)), replies, nb, rd, ret), logger)
      for (((src, p), i) <- preAccepts.zipWithIndex) {
        val back = chan[Replica[Transport]](src, Replica.serializer)
        val mine = (replies(i) & bit(index)) != 0 || (replies(k + i) & bit(index)) != 0        // processed, or answered again
        if ((replies(2 * k + i) & bit(index)) != 0)                                             // :1176-1186 Nack(instance, largestBallot)
          back.send(ReplicaInbound().withNack(Nack(p.instance, Ballot(nb(i) >> 3, nb(i) & 7))))
        else if (mine) {
          val w = rd.slice((i * n + index) * n, (i * n + index + 1) * n)
          back.send(ReplicaInbound().withPreAcceptOk(PreAcceptOk(p.instance, p.ballot, index, 0,
            prefixSet(w, p.instance.replicaIndex, ret(i * n + index), p.instance.instanceNumber))))
        } else if ((replies(3 * k + i) & bit(index)) != 0)                                      // :1228-1238 the Commit back
          engine.depsOf.get((p.instance.replicaIndex, p.instance.instanceNumber)).foreach { case (w, end) =>
            back.send(ReplicaInbound().withCommit(Commit(p.instance, engine.triples(ret(k * n + i * n + index)), 0,
                                                         prefixSet(w, p.instance.replicaIndex, end, p.instance.instanceNumber))))
          }
      }
      preAccepts.clear()
    }
    // ---- Accepts (:1421-1511) and Prepares (:1632-1757) from replicas outside, at THIS replica
    if (accepts.nonEmpty) {
      val k = accepts.size; val as = accepts.map(_._2)
      val keyset = as.map(a => if (a.commandOrNoop.value.isNoop) (-1, false) else engine.classify(a.commandOrNoop.getCommand))
      val first = engine.triples.size; as.foreach(a => engine.triples += a.commandOrNoop)
      for ((a, i) <- as.zipWithIndex) {   // the device keeps an accepted triple by its id: its dependencies stay here
        val v = a.dependencies.intPrefixSet(a.instance.replicaIndex).value
        engine.tripleDeps(first + i) = (watermarks(a.dependencies), if (v.isEmpty) 0 else v.max + 1)
      }
      val replies = new Array[Byte](4 * k); val nb = new Array[Int](k)
      Native.check(Native.epxAccept(engine.handle, k, as.map(_.instance.replicaIndex).toArray, as.map(_.instance.instanceNumber).toArray,
                                    as.map(_.ballot.ordering).toArray, as.map(_.ballot.replicaIndex).toArray, Array.tabulate(k)(first + _),
                                    keyset.map(_._1).toArray, keyset.map(x => (if (x._2) 1 else 0).toByte).toArray,
                                    Array.fill(k)(bit(index)), replies, nb), logger)
      for (((src, a), i) <- accepts.zipWithIndex) {
        val back = chan[Replica[Transport]](src, Replica.serializer)
        if ((replies(i) & bit(index)) != 0) back.send(ReplicaInbound().withAcceptOk(AcceptOk(a.instance, a.ballot, index)))
        else if ((replies(k + i) & bit(index)) != 0) back.send(ReplicaInbound().withNack(Nack(a.instance, Ballot(nb(i) >> 3, nb(i) & 7))))
      }
      accepts.clear()
    }
    if (prepares.nonEmpty) {
      val k = prepares.size; val ps = prepares.map(_._2)
      val replies = new Array[Byte](3 * k); val nb = new Array[Int](k); val ok = new Array[Int](3 * k * n)
      Native.check(Native.epxPrepare(engine.handle, k, n, ps.map(_.instance.replicaIndex).toArray, ps.map(_.instance.instanceNumber).toArray,
                                     ps.map(_.ballot.ordering).toArray, ps.map(_.ballot.replicaIndex).toArray, Array.fill(k)(bit(index)),
                                     replies, nb, ok), logger)
      for (((src, p), i) <- prepares.zipWithIndex) {
        val back = chan[Replica[Transport]](src, Replica.serializer)
        val L = p.instance.replicaIndex; val x = p.instance.instanceNumber
        if ((replies(k + i) & bit(index)) != 0) back.send(ReplicaInbound().withNack(Nack(p.instance, Ballot(nb(i) >> 3, nb(i) & 7))))
        else if ((replies(2 * k + i) & bit(index)) != 0) {
          // :1746-1756 a CommittedEntry: "No need to run the protocol" -- the Commit goes back
          val entry = new Array[Int](6 + n)
          Native.check(Native.epxReadCmdlog(engine.handle, n, index, L, x, entry), logger)
          engine.depsOf.get((L, x)).foreach { case (w, end) =>
            back.send(ReplicaInbound().withCommit(Commit(p.instance, engine.triples(entry(3)), 0, prefixSet(w, L, end, x))))
          }
        } else if ((replies(i) & bit(index)) != 0) {
          val status = ok(i * n + index); val vote = ok(k * n + i * n + index); val triple = ok(2 * k * n + i * n + index)
          // the entry's triple: its dependencies are in the command log (what THIS replica answered the PreAccept with), or
          // -- an entry an Accept wrote names its triple by id -- with the engine (tripleDeps)
          val seen = status != 0 && triple >= 0
          val deps: Option[InstancePrefixSetProto] =
            if (!seen) None
            else {
              val entry = new Array[Int](6 + n)
              Native.check(Native.epxReadCmdlog(engine.handle, n, index, L, x, entry), logger)
              val (w, end) = if (entry(5) >= 0) (entry.slice(5, 5 + n), entry(5 + n))
                             else engine.tripleDeps.getOrElse(triple, engine.depsOf.getOrElse((L, x), (Array.fill(n)(0), 0)))
              Some(prefixSet(w, L, end, x))
            }
          back.send(ReplicaInbound().withPrepareOk(PrepareOk(
            ballot = p.ballot, instance = p.instance, replicaIndex = index,
            voteBallot = if (vote < 0) Replica.nullBallot else Ballot(vote >> 3, vote & 7),       // (-1, -1), Replica.scala:256
            status = status match { case 0 | 1 => CommandStatus.NotSeen; case 2 => CommandStatus.PreAccepted; case _ => CommandStatus.Accepted },
            commandOrNoop = if (seen) Some(engine.triples(triple)) else None,
            sequenceNumber = if (seen) Some(0) else None,                                          // :1711-1737
            dependencies = deps)))
        }
      }
      prepares.clear()
    }
    queued = 0
  }

  // replicas of this process: the addresses a GpuEPaxosReplica was created for (set by the main that creates them)
  var hosted: Transport#Address => Boolean = _ => true
}
